#!/bin/bash
# kernel traces of the default bench with and without the folded LayerNorm (per-kernel A/B)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for f in 0 1; do
  export PSAM_FOLD_LN=$f
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_t$f -o t -- python3 bench.py --no-cpu-baseline --no-extras --steps 4 --warmup 1 > /dev/null 2>&1
  python3 tools/rocprof_summary.py gpurun_out/prof_t$f/t_results.db gpurun_out/r02_fold${f}_kernel_trace.md "PSAM_FOLD_LN=$f rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-extras --steps 4 --warmup 1"
  rm -rf gpurun_out/prof_t$f
done
