#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_s3
O=gpurun_out/r05_s3
timeout 300 python3 tools/r05/attn_accuracy.py > $O/attn_accuracy.log 2>&1
cat $O/attn_accuracy.log | grep -v amdgpu.ids
for f in 1 0; do
  echo "== PSAM_FUSE_GLOBAL_RELPOS=$f per-slice (batch 1)"
  PSAM_FUSE_GLOBAL_RELPOS=$f PSAM_STATS_BATCH=1 timeout 600 python3 tools/parity_stats.py 4 2>&1 | grep -v amdgpu.ids | tee -a $O/parity_stats.log
  echo "== PSAM_FUSE_GLOBAL_RELPOS=$f batched"
  PSAM_FUSE_GLOBAL_RELPOS=$f timeout 600 python3 tools/parity_stats.py 4 2>&1 | grep -v amdgpu.ids | tee -a $O/parity_stats.log
done
