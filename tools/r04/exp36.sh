#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_exp36; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o c5 -- python3 tools/config5_profile.py > $O/run.txt 2>&1
grep "config5" $O/run.txt
python3 tools/rocprof_summary.py $O/prof/c5_results.db $O/config5_kernel_trace.md "rocprofv3 --kernel-trace --stats -- python3 tools/config5_profile.py"
rm -rf $O/prof
head -30 $O/config5_kernel_trace.md
