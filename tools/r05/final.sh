#!/bin/bash
# round-5 record: full GPU test suite, smoke, default bench + kernel trace + PMC traffic (tools/final_profile.sh), per-shape GEMM traffic,
# kernel traces of configs 2 and 5.   usage: bash tools/r05/final.sh <tag>
TAG=${1:-r05_a}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 2400 python3 -m pytest tests -q -m gpu > $O/${TAG}_pytest.log 2>&1; echo "pytest rc $?" >> $O/${TAG}_pytest.log
tail -4 $O/${TAG}_pytest.log
timeout 600 python3 __graft_entry__.py smoke 2>&1 | tail -2
# the per-shape GEMM traffic first: bench.py takes `roofline.traffic` from profiles/r05_gemm_traffic_by_shape.json only when its source
# hashes match the library being run (copy the fresh file into profiles/ of the work tree afterwards as well)
bash tools/gemm_traffic_by_shape.sh 0 > /dev/null 2>&1
cp gpurun_out/gemm_traffic_by_shape.json profiles/r05_gemm_traffic_by_shape.json
bash tools/final_profile.sh $TAG > /dev/null 2>&1
python3 - <<PY
import json
j=json.load(open("gpurun_out/${TAG}_bench_default.json"))
print(j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["traffic"], j["stage_ms_per_step"])
for k in ("per_slice_forward","overlap_streams_auto","rank_of_8_strong","no_support_cache","sparse_volume"):
    print(k, {a:b for a,b in j[k].items() if a!="note"})
for k,v in j["other_configs"].items():
    print(k, v["value"], v.get("ms_per_call"))
print(j["cpu_baseline"])
print(j["parity_vs_cpu_oracle"]["worst_max_abs_dprob_low_res"], j["parity_vs_cpu_oracle"]["min_dice_final_mask"])
PY
python3 -c "
import json; d=json.load(open('gpurun_out/gemm_traffic_by_shape.json'))
for k,v in d['shapes'].items(): print(k, {a: v[a] for a in v if 'ratio' in a or 'bytes' in a})
"
for w in batched per_slice; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_c2$w -o t -- python3 tools/config2_profile.py $w > $O/${TAG}_config2_$w.log 2>&1
  DB=$(ls $O/prof_c2$w/*.db $O/prof_c2$w/*/*.db 2>/dev/null | head -1)
  python3 tools/rocprof_summary.py $DB $O/${TAG}_config2_${w}_kernel_trace.md "rocprofv3 --kernel-trace --stats -- python3 tools/config2_profile.py $w  (BASELINE config 2: 8 slices x (2 warm-up + 6 timed) calls)"
  rm -rf $O/prof_c2$w
  grep config2 $O/${TAG}_config2_$w.log
done
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_c5 -o t -- python3 tools/config5_profile.py > $O/${TAG}_config5.log 2>&1
DB=$(ls $O/prof_c5/*.db $O/prof_c5/*/*.db 2>/dev/null | head -1)
python3 tools/rocprof_summary.py $DB $O/${TAG}_config5_kernel_trace.md "rocprofv3 --kernel-trace --stats -- python3 tools/config5_profile.py  (BASELINE config 5: ProtoMedSAM.forward_classes, 10 calls)"
rm -rf $O/prof_c5
grep config5 $O/${TAG}_config5.log
