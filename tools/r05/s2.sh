#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_s2
O=gpurun_out/r05_s2
timeout 900 python3 -m pytest tests/test_kernels_core_gpu.py -x -q -k "attention" > $O/pytest_attn.log 2>&1
tail -5 $O/pytest_attn.log
timeout 300 python3 tools/r05/attn_bench.py > $O/attn_bench.log 2>&1
grep global $O/attn_bench.log
PSAM_GEMM_LOG=1 timeout 600 python3 tools/r05/gemm_small_sweep.py 0,1,15,16 > $O/gemm_small.log 2>&1
grep -v "psam_gemm_f16:" $O/gemm_small.log
timeout 900 python3 -m pytest tests/test_sam_gpu.py tests/test_fullsize_gpu.py -x -q > $O/pytest_sam.log 2>&1
tail -5 $O/pytest_sam.log
