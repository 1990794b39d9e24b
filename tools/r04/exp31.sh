#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_exp31; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_core_gpu.py -x -q -m gpu -k "attention" 2>&1 | tail -15
python tools/config3_profile.py 2>&1 | tail -1 | tee $O/c3.txt
