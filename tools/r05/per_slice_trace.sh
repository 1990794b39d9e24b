#!/bin/bash
# one ProtoSAM.forward per slice (headline model, micro-batch 1): wall time with and without the second stream, then the kernel trace of
# the single-stream form -> gpurun_out/r05_per_slice_kernel_trace.md
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 300 python3 tools/per_slice_profile.py 1 16 auto 4 2>&1 | grep "^batch"
timeout 300 python3 tools/per_slice_profile.py 1 16 0 4 2>&1 | grep "^batch"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_ps -o t -- python3 tools/per_slice_profile.py 1 16 0 3 > $O/r05_per_slice.log 2>&1
DB=$(ls $O/prof_ps/*.db $O/prof_ps/*/*.db 2>/dev/null | head -1)
python3 tools/rocprof_summary.py $DB $O/r05_per_slice_kernel_trace.md "rocprofv3 --kernel-trace --stats -- python3 tools/per_slice_profile.py 1 16 0 3  (3 setup calls + 5 x 16 one-slice ProtoSAM.forward calls, single stream)"
rm -rf $O/prof_ps
grep "^batch" $O/r05_per_slice.log
head -40 $O/r05_per_slice_kernel_trace.md | cut -c1-160
