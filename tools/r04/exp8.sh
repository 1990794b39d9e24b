#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_exp8; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_core_gpu.py -x -q -m gpu -k "window" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -3 $O/pytest.log
for rep in 1 2; do for w in 2 3; do for b in 16; do echo "PSAM_WATTN=$w"; FUSED=1 PSAM_WATTN=$w python tools/attn_win_bench.py $b 2>/dev/null; done; done; done | tee $O/win_bench.log
