#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_exp30; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o c3 -- python3 tools/config3_profile.py > $O/run.txt 2>&1
grep "config 3" $O/run.txt
python3 tools/rocprof_summary.py $O/prof/c3_results.db $O/config3_kernel_trace.md "rocprofv3 --kernel-trace --stats -- python3 tools/config3_profile.py (2 warm-up + 4 timed passes over the 32-slice volume, 16-slice batches)"
rm -rf $O/prof
head -32 $O/config3_kernel_trace.md
