#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_exp33; mkdir -p $O
TILE=15 PSAM_GEMM_ASM_CO=build/gemm_asm_exp.co timeout 300 python tools/gemm_asm_ab.py 0,36,37,0,36,37 "65536x5120x1280x1" 2>/dev/null | tee $O/gelu_bound.txt
