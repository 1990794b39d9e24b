#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
FUSED=1 timeout 240 rocprofv3 --pmc $set -d gpurun_out/pmc_w -o w -- python3 tools/attn_win_bench.py 8 > /dev/null 2>&1
python3 - <<PY
import sqlite3, glob
db = glob.glob("gpurun_out/pmc_w/*.db")[0]
cur = sqlite3.connect(db).cursor()
for r in cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%attn%' group by kernel_name, counter_name"):
    print(f"{r[1]:28s} {r[2]:16.0f} n={r[3]}")
PY
rm -rf gpurun_out/pmc_w
done
