#!/bin/bash
# round 4, sixth GPU call: whole-volume / reference-record tests, heavy-tail error printout, pre_slot + residual policy variants, other configs
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_exp6; mkdir -p $O
timeout 1500 python -m pytest tests/test_fullsize_gpu.py -x -q -s -m gpu > $O/pytest_fullsize.log 2>&1; echo "pytest rc $?" >> $O/pytest_fullsize.log
grep -E "config|passed|failed|rc " $O/pytest_fullsize.log | tail -40
PSAM_GEMM_ASM_CO=build/gemm_asm_exp.co PSAM_GEMM_ASM_TRACE=1 timeout 600 python tools/gemm_asm_ab.py 0,1,31,32,33,34,35 "65536x1280x1280x2;65536x1280x5120x2" > $O/ab_pre.log 2>&1
grep -v "^asm" $O/ab_pre.log
grep "^asm" $O/ab_pre.log | awk '{print $2,$3,$4,$5,$9,$10}' | sort | uniq -c | awk '{print $2,$3,$5,$6}' | sort | awk '{k=$1" "$2; n[k]++; a[k]+=$3; b[k]+=$4} END {for (k in n) print k, a[k]/n[k], b[k]/n[k]}' | sort
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_exp6/bench.json"))
print("headline", d["value"], d["roofline"]["achieved"], "rank_of_8", d["rank_of_8_strong"]["slices_per_s_per_rank"], "per_slice", d["per_slice_forward"]["value"], "overlap", d["overlap_streams_auto"]["value"])
for k, v in d["other_configs"].items(): print(k, v.get("value"), v.get("ms_per_call"), v.get("roofline"), v.get("error"))
PY
