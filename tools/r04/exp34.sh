#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_exp34; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_core_gpu.py -x -q -m gpu -k "gemm" 2>&1 | tail -6
TILE=15 timeout 300 python tools/gemm_asm_ab.py 0 "65536x5120x1280x1;65536x3072x768x1" 2>/dev/null | tee $O/gelu.txt
TILE=16 timeout 300 python tools/gemm_asm_ab.py 0 "4096x5120x1280x1" 2>/dev/null | tee -a $O/gelu.txt
