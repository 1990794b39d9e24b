#!/bin/bash
# full GPU suite + bench with the assembly window kernel (A/B against the HIP one)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_exp10; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
bash tools/ab_env.sh PSAM_WATTN "3 2" --no-other-configs --no-extras 2>&1 | tee $O/bench_ab.log
