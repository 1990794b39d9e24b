#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_exp23; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_core_gpu.py -x -q -m gpu -k "attention or gattn or global" 2>&1 | tail -4
for f in lazy1 bal lazy1 bal; do
ABL=$f PSAM_GEMM_ASM_CO=build/gattn/$f.co timeout 120 python tools/gattn_ablate.py 2>&1 | grep -v amdgpu.ids | tee -a $O/lazy.txt
done
