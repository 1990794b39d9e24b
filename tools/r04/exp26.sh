#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_exp26; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_core_gpu.py -x -q -m gpu -k "attention or window" 2>&1 | tail -2
ABL=heads python tools/wattn_time.py 2>&1 | grep window | tee -a $O/t.txt
ABL=heads python tools/wattn_time.py 2>&1 | grep window | tee -a $O/t.txt
NCALLS=30 bash tools/r04/exp25.sh 2>&1 | tail -3 | tee -a $O/t.txt
