#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_s11; mkdir -p $O
timeout 900 python3 -m pytest tests/test_sam_gpu.py tests/test_protosam_gpu.py -x -q -s > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
grep "split\|depth\|passed\|failed\|rc " $O/pytest.log | tail -12
for sp in 1 0; do
  echo "== PSAM_SPLIT_FP16=$sp config 4 batched / per-slice"
  PSAM_SPLIT_FP16=$sp timeout 600 python3 tools/parity_stats.py 4 2>&1 | grep -v amdgpu.ids | tee -a $O/parity_stats.log
  PSAM_SPLIT_FP16=$sp PSAM_STATS_BATCH=1 timeout 600 python3 tools/parity_stats.py 4 2>&1 | grep -v amdgpu.ids | tee -a $O/parity_stats.log
done
timeout 900 python3 bench.py --no-cpu-baseline --no-other-configs > $O/bench_split1.json 2> $O/bench1.log
PSAM_SPLIT_FP16=0 timeout 900 python3 bench.py --no-cpu-baseline --no-other-configs > $O/bench_split0.json 2> $O/bench0.log
python3 - <<'PY'
import json
for n in ("bench_split1", "bench_split0"):
    j=json.load(open(f"gpurun_out/r05_s11/{n}.json"))
    print(n, j["value"], j["ms_per_step"], j["roofline"]["frac"], j["stage_ms_per_step"], j["overlap_streams_auto"], j["per_slice_forward"]["value"])
    for r in j["roofline"]["by_shape"][:6]: print("   ", r)
PY
