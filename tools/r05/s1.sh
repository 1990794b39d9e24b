#!/bin/bash
# session 1: the new attention kernels (tests + timing), GEMM dispatch log of configs 5 / 2
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_s1
O=gpurun_out/r05_s1
timeout 900 python3 -m pytest tests/test_kernels_core_gpu.py -x -q -k "attention" > $O/pytest_attn.log 2>&1
tail -15 $O/pytest_attn.log
timeout 300 python3 tools/r05/attn_bench.py > $O/attn_bench.log 2>&1
cat $O/attn_bench.log
PSAM_GEMM_LOG=1 timeout 300 python3 tools/config5_profile.py > $O/config5.log 2>&1
tail -40 $O/config5.log
