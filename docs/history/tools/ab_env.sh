#!/bin/bash
# within-run A/B of an environment switch on the default bench: ab_env.sh VAR "v1 v2 ..." [extra bench args]
VAR=$1; VALS=$2; shift 2
for rep in 1 2; do for v in $VALS; do
  env $VAR=$v python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 > /tmp/ab_line.json
  python - "$VAR=$v" <<'PY'
import json, sys
d = json.load(open("/tmp/ab_line.json"))
r = d["roofline"]
print(sys.argv[1], "slices/s", d["value"], "ms/step", d["ms_per_step"], "gemm TF/s", r["achieved"], "gemm share", r["gemm_time_share"])
PY
done; done
