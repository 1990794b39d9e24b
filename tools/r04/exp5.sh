#!/bin/bash
# round 4, fifth GPU call: full GPU test suite (minus the records still being generated), library comparison, PMC issue counters, other configs
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_exp5; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu -k "not whole_volume and not reference_full" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -6 $O/pytest.log
timeout 900 python tools/gemm_vs_library.py 2>/dev/null | tee $O/gemm_vs_library.txt
( for sh in "65536 3840 1280 15 0 1" "65536 1280 1280 15 2 1" "65536 5120 1280 15 1 1" "65536 1280 5120 15 2 1" "65536 2304 768 15 0 1" "65536 768 768 15 2 1" "65536 3072 768 15 1 1" "65536 768 3072 15 2 1"; do
    echo "== $sh (M N K tile epilogue ln)"; bash tools/pmc_gemm.sh $sh 2>&1 | grep -E "SQ_WAVE_CYCLES|SQ_BUSY_CYCLES|SQ_VALU_MFMA_BUSY|SQ_WAIT_INST_ANY|SQ_VMEM_TA_ADDR_FIFO_FULL|TCP_PENDING_STALL|SQ_ACTIVE_INST_ANY "; done ) > $O/gemm_pmc_sq.txt 2>&1
tail -30 $O/gemm_pmc_sq.txt
python bench.py --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/bench_other.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_exp5/bench_other.json"))
print("headline", d["value"], d["roofline"]["achieved"])
for k, v in d["other_configs"].items(): print(k, {a: v.get(a) for a in ("value", "ms_per_call", "gemm_tflops", "gemm_frac_of_mfma_peak", "gemm_time_share", "error")})
PY
