#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_exp9; mkdir -p $O
( echo "shipped"; FUSED=1 python tools/attn_win_bench.py 16 2>/dev/null | head -1
for f in build/wattn_*.co; do echo $f; PSAM_GEMM_ASM_CO=$f FUSED=1 timeout 120 python tools/attn_win_bench.py 16 2>/dev/null | head -1; done
echo "shipped"; FUSED=1 python tools/attn_win_bench.py 16 2>/dev/null | head -1 ) | tee $O/ablate.log
