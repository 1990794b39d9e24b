#!/bin/bash
# tile 16 with the capacity-aware epilogue scheduler: correctness, then traced schedules (v1 = shipped, v4..v10 = other capacities)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_exp13; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_core_gpu.py -x -q -m gpu -k "gemm" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
agg() { grep "^asm" $1 | awk '{print $2,$3,$4,$5,$9,$10}' | sort | uniq -c | awk '{print $2,$3,$5,$6}' | sort | awk '{k=$1" "$2; n[k]++; a[k]+=$3; b[k]+=$4} END {for (k in n) print k, a[k]/n[k], b[k]/n[k]}' | sort -V; }
TILE=16 PSAM_GEMM_ASM_CO=build/gemm_asm_exp.co PSAM_GEMM_ASM_TRACE=1 timeout 600 python tools/gemm_asm_ab.py 0,1,2,4,5,6,7,8,9,10 "65536x5120x1280x1" > $O/ab_t16_gelu.log 2>&1
grep -v "^asm" $O/ab_t16_gelu.log; agg $O/ab_t16_gelu.log
TILE=16 PSAM_GEMM_ASM_CO=build/gemm_asm_exp.co PSAM_GEMM_ASM_TRACE=1 timeout 600 python tools/gemm_asm_ab.py 0,1,2 "65536x3840x1280x0;65536x1280x5120x2;65536x1280x1280x2" > $O/ab_t16_rest.log 2>&1
grep -v "^asm" $O/ab_t16_rest.log; agg $O/ab_t16_rest.log
TILE=15 timeout 300 python tools/gemm_asm_ab.py 0 "65536x5120x1280x1;65536x3840x1280x0;65536x1280x5120x2;65536x1280x1280x2" 2>/dev/null | tee $O/ab_t15.log
