#!/bin/bash
# round 4, second GPU call: deep residual ring (AGPR landing), dual-stream half-chip GEMM chains
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_exp2; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_core_gpu.py -x -q -m gpu -k "gemm" > $O/pytest_gemm.log 2>&1; echo "pytest rc $?" >> $O/pytest_gemm.log
tail -3 $O/pytest_gemm.log
PSAM_GEMM_ASM_CO=build/gemm_asm_exp.co PSAM_GEMM_ASM_TRACE=1 timeout 600 python tools/gemm_asm_ab.py 0,1,25 "65536x1280x1280x2;65536x1280x5120x2;65536x768x768x2;20752x768x3072x2" > $O/ab_f32.log 2>&1
grep -v "^asm" $O/ab_f32.log
grep "^asm" $O/ab_f32.log | awk '{print $2,$3,$4,$5,$9,$10}' | sort | uniq -c | awk '{print $2,$3,$5,$6}' | sort | awk '{k=$1" "$2; n[k]++; a[k]+=$3; b[k]+=$4} END {for (k in n) print k, a[k]/n[k], b[k]/n[k]}' | sort
timeout 300 python tools/gemm_ln_bench.py 2>&1 | tee $O/ln_bench.log
timeout 600 python tools/r04/dual_stream_gemm.py 8 2>&1 | tee $O/dual.log
bash tools/ab_env.sh PSAM_DUMMY "0 1" --no-other-configs --no-extras 2>&1 | tee $O/bench.log
