#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_s10; mkdir -p $O
for f in 0.8 0; do
for c in 3 4; do
  echo "== config $c per-slice, PSAM_FOLD_MIN_FILL=$f"
  PSAM_FOLD_MIN_FILL=$f PSAM_STATS_BATCH=1 timeout 600 python3 tools/parity_stats.py $c 2>&1 | grep -v amdgpu.ids | tee -a $O/parity_stats.log
done; done
echo "== tile 1 everywhere (PSAM_GEMM_TILE=1), config 3 per-slice"
PSAM_GEMM_TILE=1 PSAM_STATS_BATCH=1 timeout 600 python3 tools/parity_stats.py 3 2>&1 | grep -v amdgpu.ids | tee -a $O/parity_stats.log
echo "== HIP attention kernels (PSAM_GATTN=1 PSAM_WATTN=3), config 3 per-slice"
PSAM_GATTN=1 PSAM_WATTN=3 PSAM_STATS_BATCH=1 timeout 600 python3 tools/parity_stats.py 3 2>&1 | grep -v amdgpu.ids | tee -a $O/parity_stats.log
