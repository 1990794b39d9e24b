#!/bin/bash
# round 4, fourth GPU call: mixed-support batches, dual-stream SAM encoder (tests + bench legs)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_exp4; mkdir -p $O
timeout 1200 python -m pytest tests/test_protosam_gpu.py tests/test_sam_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -15 $O/pytest.log
for rep in 1 2; do for d in 0 auto; do
  PSAM_SAM_DUAL=$d python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 > /tmp/l.json
  python - $d <<'PY'
import json, sys
d = json.load(open("/tmp/l.json")); r = d["roofline"]
print("dual", sys.argv[1], "headline", d["value"], "gemm TF/s", r["achieved"], "| overlap_auto", d["overlap_streams_auto"]["value"], "| per_slice", d["per_slice_forward"]["value"],
      "| rank_of_8", d["rank_of_8_strong"]["slices_per_s_per_rank"], "| sparse", d["sparse_volume"]["value"], "| no_cache", d["no_support_cache"]["value"])
PY
done; done 2>&1 | tee $O/bench_dual.log
