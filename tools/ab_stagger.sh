#!/bin/bash
for st in ${1:-0 32 64 128}; do echo "== stagger $st"; PSAM_GEMM_STAGGER=$st python3 tools/gemm_ksweep.py 10 65536 3840 | grep "K= 1280"; PSAM_GEMM_STAGGER=$st python3 tools/gemm_ksweep.py 10 65536 1280 f32 | grep "K= 1280\|K= 5120"; done
