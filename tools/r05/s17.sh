#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_s17; mkdir -p $O
timeout 900 python3 -m pytest tests/test_protosam_gpu.py tests/test_kernels_core_gpu.py -x -q > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
tail -4 $O/pytest.log
timeout 900 python3 bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.log
python3 - <<'PY'
import json
j=json.load(open("gpurun_out/r05_s17/bench.json"))
print(j["value"], j["ms_per_step"], j["roofline"]["frac"], j["stage_ms_per_step"], j["per_slice_forward"]["value"], j["overlap_streams_auto"]["value"], j["overlap_streams_auto"]["single_stream_same_minute"])
for k,v in j["other_configs"].items():
    print(k, v["value"], v.get("ms_per_call"))
PY
