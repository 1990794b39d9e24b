#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_s18; mkdir -p $O
timeout 2400 python3 -m pytest tests/test_fullsize_gpu.py -q -s -k "whole_volume and not 4-1234-0 and not 4-4242-0" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
grep "weights\|slices whose\|passed\|failed\|^E  .*Assert" $O/pytest.log | cut -c1-330
