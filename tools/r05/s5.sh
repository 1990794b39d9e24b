#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_s6
O=gpurun_out/r05_s6
timeout 900 python3 -m pytest tests/test_kernels_core_gpu.py -x -q -k "gemm_deep or test_gemm\[" > $O/pytest_gemm.log 2>&1
tail -4 $O/pytest_gemm.log
timeout 600 python3 tools/r05/gemm_small_sweep.py 0,1,12,13,16 2>&1 | grep -v amdgpu | tee $O/gemm_small.log
timeout 900 python3 bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench.log
python3 - <<'PY'
import json
j=json.load(open("gpurun_out/r05_s6/bench_default.json"))
print(j["value"], j["ms_per_step"], j["roofline"]["frac"], j["stage_ms_per_step"])
for k in ("per_slice_forward","overlap_streams_auto","rank_of_8_strong","no_support_cache","sparse_volume"):
    print(k, {a:b for a,b in j[k].items() if a!="note"})
for k,v in j["other_configs"].items():
    print(k, v["value"], v.get("ms_per_call"))
PY
