#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 2400 python3 -m pytest tests/test_fullsize_gpu.py -q -s -k "whole_volume" > $O/r05_whole_volume.log 2>&1; echo "pytest rc $?" >> $O/r05_whole_volume.log
grep "weights\|slices whose\|passed\|failed\|rc " $O/r05_whole_volume.log | cut -c1-330
