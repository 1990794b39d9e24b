#!/bin/bash
for st in ${1:-0 32 64 128}; do echo "== stagger $st"; PSAM_GEMM_STAGGER=$st python3 tools/gemm_tiles.py ${2:-11} "78400x3840x1280;65536x5120x1280;65536x3840x1280"; done
