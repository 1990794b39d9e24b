#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_s4
O=gpurun_out/r05_s4
timeout 900 python3 -m pytest tests/test_kernels_core_gpu.py -x -q -k "attention" > $O/pytest_attn.log 2>&1
tail -3 $O/pytest_attn.log
timeout 300 python3 tools/r05/attn_bench.py > $O/attn_bench.log 2>&1
cat $O/attn_bench.log | grep -v amdgpu
timeout 300 python3 tools/config5_profile.py 2>&1 | grep -v amdgpu | tee $O/config5.log
timeout 900 python3 bench.py > $O/bench_default.json 2> $O/bench.log
python3 - <<'PY'
import json
j=json.load(open("gpurun_out/r05_s4/bench_default.json"))
print(j["value"], j["ms_per_step"], j["roofline"]["frac"], j["stage_ms_per_step"])
for k in ("per_slice_forward","overlap_streams_auto","rank_of_8_strong","no_support_cache","sparse_volume"):
    print(k, {a:b for a,b in j[k].items() if a!="note"})
for k,v in j["other_configs"].items():
    print(k, v["value"], v.get("ms_per_call"))
print(j["parity_vs_cpu_oracle"])
PY
