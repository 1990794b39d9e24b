#!/bin/bash
# SQ counters for the global-attention microbench (one --pmc pass, bounded by timeout)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d gpurun_out/pmc_a -o a -- python3 tools/attn_glob_bench.py ${1:-4} > /dev/null 2>&1
python3 - <<PY
import sqlite3, glob
db = glob.glob("gpurun_out/pmc_a/*.db")[0]
cur = sqlite3.connect(db).cursor()
for r in cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%attn%' group by kernel_name, counter_name"):
    print(f"{r[0][:40]:40s} {r[1]:28s} {r[2]:16.0f} n={r[3]}")
PY
rm -rf gpurun_out/pmc_a
