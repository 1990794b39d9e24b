#!/bin/bash
# kernel traces of configs 2 (batched, per slice) and 5
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_s7; mkdir -p $O
for w in batched per_slice; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_c2$w -o t -- python3 tools/config2_profile.py $w > $O/config2_$w.log 2>&1
  DB=$(ls $O/prof_c2$w/*.db $O/prof_c2$w/*/*.db 2>/dev/null | head -1)
  python3 tools/rocprof_summary.py $DB $O/config2_${w}_kernel_trace.md "rocprofv3 --kernel-trace --stats -- python3 tools/config2_profile.py $w  (BASELINE config 2: 8 slices x (2 warm-up + 6 timed) calls)"
  rm -rf $O/prof_c2$w
  grep config2 $O/config2_$w.log; head -30 $O/config2_${w}_kernel_trace.md | cut -c1-150
done
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_c5 -o t -- python3 tools/config5_profile.py > $O/config5.log 2>&1
DB=$(ls $O/prof_c5/*.db $O/prof_c5/*/*.db 2>/dev/null | head -1)
python3 tools/rocprof_summary.py $DB $O/config5_kernel_trace.md "rocprofv3 --kernel-trace --stats -- python3 tools/config5_profile.py  (BASELINE config 5: ProtoMedSAM.forward_classes, 10 calls)"
rm -rf $O/prof_c5
grep config5 $O/config5.log; head -30 $O/config5_kernel_trace.md | cut -c1-150
