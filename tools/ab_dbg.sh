#!/bin/bash
for d in ${1:-0 8 32}; do echo "== dbg $d"; PSAM_GEMM_DBG=$d python3 tools/gemm_ksweep.py 10 65536 3840 | grep "K=   64\|K= 1280\|fit"; PSAM_GEMM_DBG=$d python3 tools/gemm_ksweep.py 10 65536 1280 f32 | grep "K=   64\|K= 1280\|fit"; done
