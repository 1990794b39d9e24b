#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_s16; mkdir -p $O
for rep in 1 2; do
timeout 300 python3 tools/r05/attn_bench.py 2>&1 | grep "global\|B=16 N=1297\|B=1 N=5330" | sed 's/^/shipped  /' | tee -a $O/attn_dmalate.log
PSAM_GEMM_ASM_CO=build/gattn_dmalate.co timeout 300 python3 tools/r05/attn_bench.py 2>&1 | grep "global\|B=16 N=1297\|B=1 N=5330" | sed 's/^/dma-late /' | tee -a $O/attn_dmalate.log
done
PSAM_GEMM_ASM_CO=build/gattn_dmalate.co timeout 600 python3 -m pytest tests/test_kernels_core_gpu.py -q -k "attention_global" 2>&1 | tail -2
