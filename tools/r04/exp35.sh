#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_exp35; mkdir -p $O
for i in 1 2 3; do for f in old new; do
echo -n "$f " | tee -a $O/ab.txt
TILE=15 PSAM_GEMM_ASM_CO=build/ab/$f.co timeout 300 python tools/gemm_asm_ab.py 0 "65536x5120x1280x1" 2>/dev/null | tee -a $O/ab.txt
done; done
