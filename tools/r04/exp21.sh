#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_exp21; mkdir -p $O
timeout 3300 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -12 $O/pytest.log
