for sh in 8192,8192,8192 32768,3840,1280; do
for d in 0 1 2 4 8 3 5 6 7 16 15; do SHAPE=$sh TILE=7 PSAM_GEMM_DBG=$d python tools/gemm_ablate.py 2>&1 | tail -1; done; done
