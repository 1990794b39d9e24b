#!/bin/bash
# one ProtoSAM.forward per slice with the library defaults (second stream): where the GPU idles -> gpurun_out/r05_per_slice_gaps.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace -d $O/prof_pg -o t -- python3 tools/per_slice_profile.py 1 16 ${1:-auto} 4 > $O/r05_per_slice_gaps.log 2>&1
DB=$(ls $O/prof_pg/*.db $O/prof_pg/*/*.db 2>/dev/null | head -1)
python3 tools/r05/gap_report.py $DB 0.05 $O/r05_per_slice_gaps.txt
python3 tools/busy_share.py $DB 0.05 | head -3
rm -rf $O/prof_pg
grep "^batch" $O/r05_per_slice_gaps.log
