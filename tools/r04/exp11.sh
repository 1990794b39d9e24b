#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_exp11; mkdir -p $O
TILE=16 PSAM_GEMM_ASM_CO=build/gemm_asm_exp.co PSAM_GEMM_ASM_TRACE=1 timeout 600 python tools/gemm_asm_ab.py 0,1,2,3,4,5 "65536x5120x1280x1;65536x3840x1280x0;65536x1280x5120x2" > $O/ab_t16.log 2>&1
grep -v "^asm" $O/ab_t16.log
grep "^asm" $O/ab_t16.log | awk '{print $2,$3,$4,$5,$9,$10}' | sort | uniq -c | awk '{print $2,$3,$5,$6}' | sort | awk '{k=$1" "$2; n[k]++; a[k]+=$3; b[k]+=$4} END {for (k in n) print k, a[k]/n[k], b[k]/n[k]}' | sort
TILE=15 PSAM_GEMM_ASM_CO=build/gemm_asm_exp.co timeout 300 python tools/gemm_asm_ab.py 0 "65536x5120x1280x1;65536x3840x1280x0;65536x1280x5120x2" 2>/dev/null | tee $O/ab_t15.log
