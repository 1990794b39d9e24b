#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_exp24; mkdir -p $O
for f in lazy0 lazy1 lazy0 lazy1; do
ABL=$f PSAM_GEMM_ASM_CO=build/wattn/$f.co timeout 120 python tools/wattn_time.py 2>&1 | grep window | tee -a $O/lazy.txt
done
