cd $GRAFT_REPO_ROOT
timeout 600 python3 -m pytest tests/test_kernels_core_gpu.py -q -k "attention_global" 2>&1 | tail -2
bash tools/r05/gattn_ablate.sh
