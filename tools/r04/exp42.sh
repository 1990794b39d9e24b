#!/bin/bash
cd $GRAFT_REPO_ROOT
for T in 15 16; do echo "== tile $T"; TILE=$T timeout 300 python tools/gemm_asm_ab.py 0 "65536x768x768x2;65536x768x3072x2;65536x2304x768x0;65536x3072x768x1;20752x768x768x2;20752x768x3072x2;20752x2304x768x0;20752x3072x768x1" 2>/dev/null; done
