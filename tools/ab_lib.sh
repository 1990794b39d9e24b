# A/B of library builds on one box: current (A), pre-LN-fold gemm.hip (B), r02_c-era gemm.hip (C); fold off everywhere
for rep in 1 2; do
for v in A B C; do
  if [ $v = A ]; then unset PSAM_LIB_PATH; else export PSAM_LIB_PATH=$PWD/gpurun_scratch/ab/libprotosam_hip_$v.so; fi
  PSAM_FOLD_LN=0 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib $v fold 0:', d['value'], d['roofline']['achieved'], d['roofline']['gemm_time_share'])"
done
unset PSAM_LIB_PATH
PSAM_FOLD_LN=1 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib A fold 1:', d['value'], d['roofline']['achieved'], d['roofline']['gemm_time_share'])"
done
