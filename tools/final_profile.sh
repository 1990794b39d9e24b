#!/bin/bash
# Round record: default bench (JSON line), rocprofv3 kernel trace of the same command, and the two PMC passes for HBM-side
# traffic of the GEMM launches. usage: tools/final_profile.sh <tag>     (outputs under gpurun_out/, copy into profiles/)
TAG=$1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python3 bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench.log
# (--no-extras: only the setup, warm-up and timed steps of the headline configuration are in the trace, so that the per-kernel
# averages can be compared with the bench line's roofline object)
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_t -o t -- python3 bench.py --no-cpu-baseline --no-extras > gpurun_out/${TAG}_bench_under_rocprof.json 2>/dev/null
python3 tools/rocprof_summary.py gpurun_out/prof_t/t_results.db gpurun_out/${TAG}_kernel_trace.md "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-extras  (3 one-slice setup calls + 2 warm-up + 6 timed steps of 16 slices)"
rm -rf gpurun_out/prof_t
timeout 900 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_f -o f -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_w -o w -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2>&1
F=$(ls gpurun_out/pmc_f/*.db gpurun_out/pmc_f/*/*.db 2>/dev/null | head -1); W=$(ls gpurun_out/pmc_w/*.db gpurun_out/pmc_w/*/*.db 2>/dev/null | head -1)
python3 tools/pmc_traffic.py $F $W gpurun_out/${TAG}_gemm_pmc_traffic.json "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1"
rm -rf gpurun_out/pmc_f gpurun_out/pmc_w
cat gpurun_out/${TAG}_bench_default.json
