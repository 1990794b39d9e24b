#!/bin/bash
# one-slice shapes (M = 4096): cycles inside the k-loop / epilogue vs the launch's wall time, tiles 15 and 16
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_exp17; mkdir -p $O
SH="4096x3840x1280x0;4096x1280x1280x2;4096x5120x1280x1;4096x4096x1280x1;4096x1024x1280x1;4096x1280x5120x2"
for T in 15 16; do
TILE=$T PSAM_GEMM_ASM_CO=build/gemm_asm_exp.co PSAM_GEMM_ASM_TRACE=1 timeout 300 python tools/gemm_asm_ab.py 0,1 "$SH" > $O/t$T.log 2>&1
grep -v "^asm" $O/t$T.log | grep -v amdgpu.ids
grep "^asm" $O/t$T.log | sort | uniq -c | sort -k4 | awk '{$1=""; print}' | sort -u
done
