cd $GRAFT_REPO_ROOT
for rep in 1 2; do
timeout 300 python3 tools/r05/attn_bench.py 2>&1 | grep "global" | sed 's/^/phase2 (shipped) /'
PSAM_GEMM_ASM_CO=build/gattn_rh_phase1.co timeout 300 python3 tools/r05/attn_bench.py 2>&1 | grep "global" | sed 's/^/phase1           /'
done
