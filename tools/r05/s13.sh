#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_s13; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o t -- python3 tools/r05/decoder_bench.py > $O/dec.log 2>&1
DB=$(ls $O/prof/*.db $O/prof/*/*.db 2>/dev/null | head -1)
python3 tools/rocprof_summary.py $DB $O/decoder_kernel_trace.md "rocprofv3 --kernel-trace --stats -- python3 tools/r05/decoder_bench.py (13 calls of predict_masks_tokens, 27 prompt sets)"
rm -rf $O/prof
head -40 $O/decoder_kernel_trace.md | cut -c1-160
