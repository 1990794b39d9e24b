#!/bin/bash
# HBM-side traffic of the window / global attention kernels (PMC, one counter per pass; FETCH_SIZE in KiB, x2 per the gfx950 correction for wide reads)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_exp25; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
timeout 240 rocprofv3 --pmc $c -d $O/pmc_$c -o w -- python3 tools/wattn_time.py > /dev/null 2>&1
python3 - <<PY | tee -a $O/traffic.txt
import sqlite3, glob
db = (glob.glob("$O/pmc_$c/*.db") + glob.glob("$O/pmc_$c/*/*.db"))[0]
cur = sqlite3.connect(db).cursor()
for r in cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%attn%' group by kernel_name, counter_name"):
    print(f"{r[0][:40]:40s} {r[1]:12s} {r[2]:14.0f} KiB per launch  n={r[3]}")
PY
rm -rf $O/pmc_$c
done
echo "algorithmic: qkv 16*4096*3840*2 = 503.3 MB read, out 16*4096*1280*2 = 167.8 MB written" | tee -a $O/traffic.txt
