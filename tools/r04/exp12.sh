#!/bin/bash
# tile 16: where do extra VALU instructions fit in the k-loop? (filler variants v4..v15 of gemm_asm2_gen.py, no epilogue, traced)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_exp12; mkdir -p $O
TILE=16 PSAM_GEMM_ASM_CO=build/gemm_asm_exp.co PSAM_GEMM_ASM_TRACE=1 timeout 600 python tools/gemm_asm_ab.py 2,3,4,5,6,7,8,9,10,11,12,13,14,15 "65536x5120x1280x0" > $O/ab_t16.log 2>&1
grep -v "^asm" $O/ab_t16.log
grep "^asm" $O/ab_t16.log | awk '{print $2,$3,$4,$5,$9,$10}' | sort | uniq -c | awk '{print $2,$3,$5,$6}' | sort | awk '{k=$1" "$2; n[k]++; a[k]+=$3; b[k]+=$4} END {for (k in n) print k, a[k]/n[k], b[k]/n[k]}' | sort -V
