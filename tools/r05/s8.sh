#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_s8; mkdir -p $O
timeout 2400 python3 -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -25 $O/pytest.log
grep "^config" $O/pytest.log | head -40
