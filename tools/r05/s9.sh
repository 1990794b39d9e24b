#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_s9; mkdir -p $O
timeout 2400 python3 -m pytest tests/test_fullsize_gpu.py -q -k whole_volume -s > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
grep "^config\|passed\|failed" $O/pytest.log | head -40
