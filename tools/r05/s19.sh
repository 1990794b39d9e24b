#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_s19; mkdir -p $O
PSAM_SPLIT_FP16=0 timeout 2400 python3 -m pytest tests/test_fullsize_gpu.py -q -s -k "whole_volume and not 4-1234-0 and not 4-4242-0" > $O/pytest_nosplit.log 2>&1; echo "pytest rc $?" >> $O/pytest_nosplit.log
grep "weights\|passed\|failed\|Assert" $O/pytest_nosplit.log | cut -c1-330
