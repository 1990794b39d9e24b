#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_exp22; mkdir -p $O
for f in none nodma nobar nowait nodma_nobar_nowait noexp nosoft1 nosoft2 nosoft1_nosoft2 norw noreads nodeps nomfma nomfma_noreads nodma_nobar_nowait_noreads nodma_nobar_nowait_nodeps nodma_nobar_nowait_noreads_nodeps; do
ABL=$f PSAM_GEMM_ASM_CO=build/gattn/$f.co timeout 120 python tools/gattn_ablate.py 2>&1 | grep -v amdgpu.ids | tee -a $O/ablate.txt
done
