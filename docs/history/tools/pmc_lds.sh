#!/bin/bash
# per-kernel LDS bank-conflict share over one bench step (one --pmc pass)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES -d gpurun_out/pmc_l -o l -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 - <<PY
import sqlite3, glob
db = glob.glob("gpurun_out/pmc_l/*.db")[0]
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
d = {}
for k, c, v, n in rows: d.setdefault(k, {})[c] = v; d[k]["n"] = n
tot = sum(x.get("SQ_BUSY_CYCLES", 0) for x in d.values())
print(f"{'kernel':60s} {'busy %':>7s} {'lds act/busy':>12s} {'conflict/act':>12s}")
for k, x in sorted(d.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0))[:16]:
    act, conf, busy = x.get("SQ_LDS_IDX_ACTIVE", 0), x.get("SQ_LDS_BANK_CONFLICT", 0), x.get("SQ_BUSY_CYCLES", 1)
    print(f"{k[:60]:60s} {100*busy/tot:7.1f} {act/busy/8:12.2f} {conf/max(act,1):12.2f}")
PY
rm -rf gpurun_out/pmc_l
