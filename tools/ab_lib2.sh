#!/bin/bash
# A/B of two library builds on one box: the in-tree library (A) against gpurun_scratch/ab/libprotosam_hip_B.so (B)
for rep in 1 2 3; do
for v in A B; do
  if [ $v = A ]; then unset PSAM_LIB_PATH; else export PSAM_LIB_PATH=$PWD/gpurun_scratch/ab/libprotosam_hip_B.so; fi
  python bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib $v:', d['value'], 'slices/s  gemm', d['roofline']['achieved'], 'TF  share', d['roofline']['gemm_time_share'])"
done
done
unset PSAM_LIB_PATH
