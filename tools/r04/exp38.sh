#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_exp38; mkdir -p $O
for m in full coarse; do
python3 tools/per_slice_vitb_profile.py $m 2>&1 | tail -1
rocprofv3 --kernel-trace --stats -d $O/prof_$m -o ps -- python3 tools/per_slice_vitb_profile.py $m > $O/run_$m.txt 2>&1
python3 tools/busy_share.py $O/prof_$m/ps_results.db 0.2 $O/busy_$m.txt | head -3
rm -rf $O/prof_$m
done
