#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_exp16; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/prof -o ps -- python3 tools/per_slice_profile.py 1 16 0 3 > $O/prof_run.txt 2>&1
grep "slices/s" $O/prof_run.txt
python3 tools/busy_share.py $O/prof/ps_results.db 0.25 $O/busy_single_stream.txt
rocprofv3 --kernel-trace --stats -d $O/prof2 -o ps -- python3 tools/per_slice_profile.py 1 16 auto 3 > $O/prof_run2.txt 2>&1
grep "slices/s" $O/prof_run2.txt
python3 tools/busy_share.py $O/prof2/ps_results.db 0.25 $O/busy_two_streams.txt
rm -rf $O/prof $O/prof2
