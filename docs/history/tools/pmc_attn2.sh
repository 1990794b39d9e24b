#!/bin/bash
# second SQ counter set for the global-attention microbench (waits / issue mix)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM -d gpurun_out/pmc_a -o a -- python3 tools/attn_glob_bench.py ${1:-4} > /dev/null 2>&1
python3 - <<PY
import sqlite3, glob
db = glob.glob("gpurun_out/pmc_a/*.db")[0]
cur = sqlite3.connect(db).cursor()
for r in cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%attn%' group by kernel_name, counter_name"):
    print(f"{r[0][:40]:40s} {r[1]:28s} {r[2]:16.0f} n={r[3]}")
PY
rm -rf gpurun_out/pmc_a
