#!/bin/bash
# round 4, first GPU call: racc GEMM form - tests, wall A/B, traced ablations, bench A/B, K = 768 shapes
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_exp1; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_core_gpu.py -x -q -m gpu -k "gemm" > $O/pytest_gemm.log 2>&1; echo "pytest rc $?" >> $O/pytest_gemm.log
tail -5 $O/pytest_gemm.log
for r in 0 1; do PSAM_GEMM_RACC=$r timeout 300 python tools/gemm_ln_bench.py 2>&1 | sed "s/^/racc=$r /"; done | tee $O/ln_bench.log
PSAM_GEMM_ASM_CO=build/gemm_asm_exp.co PSAM_GEMM_ASM_TRACE=1 timeout 600 python tools/gemm_asm_ab.py 0,1,25,26,27 "65536x1280x1280x2;65536x1280x5120x2" > $O/ab_f32.log 2>&1
cat $O/ab_f32.log | tail -40
PSAM_GEMM_ASM_CO=build/gemm_asm_exp.co PSAM_GEMM_ASM_TRACE=1 timeout 600 python tools/gemm_asm_ab.py 0,1 "65536x2304x768x0;65536x3072x768x1;65536x768x3072x2;65536x768x768x2;20752x2304x768x0;20752x3072x768x1;20752x768x3072x2;20752x768x768x2" > $O/ab_k768.log 2>&1
tail -30 $O/ab_k768.log
bash tools/ab_env.sh PSAM_GEMM_RACC "0 1" --no-other-configs --no-extras 2>&1 | tee $O/bench_ab.log
