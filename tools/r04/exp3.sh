#!/bin/bash
# round 4, third GPU call: start-time stagger variants; cost of the per-launch events in the timed region
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_exp3; mkdir -p $O
PSAM_GEMM_ASM_CO=build/gemm_asm_exp.co timeout 900 python tools/gemm_asm_ab.py 0,26,27,28,29,30 "65536x1280x1280x2;65536x1280x5120x2;65536x5120x1280x1;65536x3840x1280x0" > $O/ab_stagger.log 2>&1
grep -v "^asm" $O/ab_stagger.log
for rep in 1 2; do
for m in torch device off; do
  if [ $m = off ]; then export PSAM_BENCH_GEMM_TIMER=0; else export PSAM_BENCH_GEMM_TIMER=1; export PSAM_TIMER_EVENTS=$m; fi
  python bench.py --no-cpu-baseline --no-other-configs --no-extras 2>/dev/null | tail -1 > /tmp/l.json
  python - $m <<'PY'
import json, sys
d = json.load(open("/tmp/l.json")); r = d["roofline"]
print("timer", sys.argv[1], "slices/s", d["value"], "ms/step", d["ms_per_step"], "gemm TF/s", r["achieved"], "share", r["gemm_time_share"], "launches", r["launches"])
PY
done; done 2>&1 | tee $O/timer_ab.log
