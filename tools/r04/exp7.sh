#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_exp7; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_core_gpu.py -x -q -m gpu -k "window" -s > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -30 $O/pytest.log
