#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels_core_gpu.py -x -q -m gpu -k "attention" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_reference_records_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])"
