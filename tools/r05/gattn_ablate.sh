#!/bin/bash
# cycles per key tile of the global rel-pos attention kernels and of their ablation builds (PSAM_GEN_GATTN_ABLATE=<x> python3 gemm_asm_gen.py
# -> build/ga_<x>.co; results of the ablated builds are wrong by construction) -> gpurun_out/r05_gattn_ablation_cycles.txt
cd $GRAFT_REPO_ROOT
for mode in fused rel; do
  for abl in none nobar nodma nobar_nodma_nowait nosoft1 nosoft2 noexp nomfma; do
    MODE=$mode ABL=$abl NCALLS=200 PSAM_GEMM_ASM_CO=build/ga_$abl.co timeout 120 python3 tools/gattn_ablate.py 2>&1 | grep -v amdgpu
  done
done | tee gpurun_out/r05_gattn_ablation_cycles.txt
