#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_exp19; mkdir -p $O
for ns in 1 0 1 0; do
PSAM_GEMM_NSPLIT=$ns python3 tools/per_slice_profile.py 1 16 auto 5 2>&1 | tail -1 | sed "s/^/nsplit $ns: /" | tee -a $O/wall.txt
done
