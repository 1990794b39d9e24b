#!/bin/bash
cd $GRAFT_REPO_ROOT
for T in 1 16 15; do echo "== tile $T"; TILE=$T timeout 300 python tools/gemm_asm_ab.py 0 "5330x768x768x2;5330x768x3072x2;1297x768x768x2;1297x768x3072x2;5330x2304x768x0;1297x2304x768x0" 2>/dev/null; done
