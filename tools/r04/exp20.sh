#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_exp20; mkdir -p $O
timeout 3000 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -6 $O/pytest.log
timeout 900 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
