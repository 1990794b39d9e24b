#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_s15; mkdir -p $O
for rep in 1 2; do
  timeout 300 python3 tools/r05/gemm_shapes_time.py 2>&1 | grep -v amdgpu | tee -a $O/gemm_dma_policy.log
  for v in nt_w nt_a nt_aw; do
    PSAM_GEMM_ASM_CO=build/gemm_$v.co timeout 300 python3 tools/r05/gemm_shapes_time.py 2>&1 | grep -v amdgpu | tee -a $O/gemm_dma_policy.log
  done
done
timeout 600 python3 tools/r05/gemm_energy_vs_library.py 2>&1 | grep -v amdgpu | tee $O/gemm_energy_vs_library.txt
