#!/bin/bash
# per-kernel L2<->fabric traffic of one bench run (PMC FETCH_SIZE / WRITE_SIZE passes), every kernel
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_exp27; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
timeout 900 rocprofv3 --pmc $c -d $O/pmc_$c -o w -- python3 bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 > /dev/null 2>&1
python3 - <<PY | tee $O/traffic_$c.txt
import sqlite3, glob
db = (glob.glob("$O/pmc_$c/*.db") + glob.glob("$O/pmc_$c/*/*.db"))[0]
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select kernel_name, count(*), sum(value), max(value) from counters_collection where counter_name = '$c' group by kernel_name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print("$c: total %.1f GB (raw KiB counters; FETCH_SIZE of wide reads counts half on gfx950)" % (tot / 1e6 * 1.024))
for r in rows[:28]:
    print(f"{r[0][:60]:60s} n={r[1]:5d}  sum {r[2] * 1.024e-3:9.1f} MB  max {r[3] * 1.024e-3:8.1f} MB  {100 * r[2] / tot:5.1f} %")
PY
rm -rf $O/pmc_$c
done
