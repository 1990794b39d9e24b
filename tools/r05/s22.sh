#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 900 python3 -m pytest tests/test_kernels_core_gpu.py tests/test_protosam_gpu.py -q -k "persistent_tile or mixed_support" 2>&1 | tail -3
timeout 900 python3 bench.py > $O/r05_b_bench_default.json 2> $O/r05_b_bench.log
python3 - <<'PY'
import json
j=json.load(open("gpurun_out/r05_b_bench_default.json"))
print(j["value"], j["ms_per_step"], j["ms_per_step_std"], j["roofline"]["frac"], j["roofline"]["achieved"], j["roofline"]["traffic"], j["power_clock"])
print(j["stage_ms_per_step"])
for r in j["roofline_hbm"]: print({k: r[k] for k in r if k in ("kernel","bound","frac","avg_launch_us","mfma_frac","achieved")})
for k in ("per_slice_forward","overlap_streams_auto","rank_of_8_strong","no_support_cache","sparse_volume"):
    print(k, {a:b for a,b in j[k].items() if a!="note"})
for k,v in j["other_configs"].items():
    print(k, v["value"], v.get("ms_per_call"), v["roofline"]["frac"])
print(j["cpu_baseline"]["value"], j["cpu_baseline"]["one_thread"]["value"], j["parity_vs_cpu_oracle"]["worst_max_abs_dprob_low_res"])
PY
