#!/bin/bash
# tile 16: what in the hidden fp16 epilogue costs cycles? (ablations v11..v16, timing only)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_exp14; mkdir -p $O
agg() { grep "^asm" $1 | awk '{print $2,$3,$4,$5,$9,$10}' | sort | uniq -c | awk '{print $2,$3,$5,$6}' | sort | awk '{k=$1" "$2; n[k]++; a[k]+=$3; b[k]+=$4} END {for (k in n) print k, a[k]/n[k], b[k]/n[k]}' | sort -V; }
TILE=16 PSAM_GEMM_ASM_CO=build/gemm_asm_exp.co PSAM_GEMM_ASM_TRACE=1 timeout 600 python tools/gemm_asm_ab.py 1,2,11,12,13,14,15,16 "65536x3840x1280x0;65536x5120x1280x1" > $O/ab.log 2>&1
grep -v "^asm" $O/ab.log; agg $O/ab.log
