#!/bin/bash
# global attention: the exp2 / packing of key block 3 deferred to the next iteration's phase 1 (shipped) against builds with no deferral /
# blocks 2,3 / block 2 (build/gl_<v>.co: PSAM_GEN_GATTN_LATE=<list> python3 gemm_asm_gen.py). Correctness of the shipped build first.
cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_kernels_core_gpu.py -q -k "attention_global" 2>&1 | tail -3
for rep in 1 2; do
for v in shipped none 23 2; do
  echo "== $v"
  if [ $v = shipped ]; then timeout 300 python3 tools/r05/attn_bench.py 2>&1 | grep -v amdgpu | grep "fused\|norel B=16\|norel B=1 N=5330"
  else PSAM_GEMM_ASM_CO=build/gl_$v.co timeout 300 python3 tools/r05/attn_bench.py 2>&1 | grep -v amdgpu | grep "fused\|norel B=16\|norel B=1 N=5330"; fi
done; done
