#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_exp18; mkdir -p $O
for T in 15 16; do
PSAM_GEMM_ASM_CO=build/gemm_asm_exp.co PSAM_GEMM_ASM_TRACE=1 timeout 300 python tools/gemm_launch_anatomy.py $T 2>&1 | grep -v amdgpu.ids | tee -a $O/anatomy.txt
done
