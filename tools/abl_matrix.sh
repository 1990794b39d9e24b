for d in 0 1 2 4 3 5 6 7 16 22 23; do PSAM_GEMM_DBG=$d TILE=7 SHAPE=8192,8192,8192 python tools/gemm_ablate.py; done
TILE=11 SHAPE=8192,8192,8192 python tools/gemm_ablate.py
TILE=11 SHAPE=65536,3840,1280 python tools/gemm_ablate.py
TILE=11 SHAPE=65536,5120,1280 python tools/gemm_ablate.py
