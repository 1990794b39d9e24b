#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_s20; mkdir -p $O
timeout 900 python3 -m pytest tests/test_kernels_core_gpu.py -x -q -k "gemm" > $O/pytest_gemm.log 2>&1; echo "rc $?" >> $O/pytest_gemm.log
tail -3 $O/pytest_gemm.log
timeout 300 python3 tools/gelu_check.py 2>&1 | grep -v amdgpu | tee $O/gelu_check.log
for rep in 1 2; do
  timeout 300 python3 tools/r05/gemm_shapes_time.py 2>&1 | grep -v amdgpu | tee -a $O/gelu_ab.log
  PSAM_GEMM_ASM_CO=build/gemm_erf.co timeout 300 python3 tools/r05/gemm_shapes_time.py 2>&1 | grep -v amdgpu | tee -a $O/gelu_ab.log
done
timeout 2400 python3 -m pytest tests/test_fullsize_gpu.py -q -s -k "whole_volume and not 4-1234-0" > $O/pytest_vol.log 2>&1; echo "pytest rc $?" >> $O/pytest_vol.log
grep "weights\|passed\|failed\|Assert" $O/pytest_vol.log | cut -c1-330
