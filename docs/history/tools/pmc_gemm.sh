#!/bin/bash
# PMC passes on one GEMM shape/tile (each counter set in its own rocprofv3 run, --pmc only). usage: pmc_gemm.sh M N K TILE [EPILOGUE 0/1/2] [LN 0/1: the folded-LayerNorm form]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
M=$1; N=$2; K=$3; T=$4; E=${5:-0}; LNF=${6:-0}
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM" \
           "TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum"; do   # (a TCC_* pass hung a box once: left out)
  i=$((i+1))
  timeout 180 rocprofv3 --pmc $set -d gpurun_out/pmc_g$i -o p -- python3 tools/gemm_one.py $M $N $K $T 3 $E $LNF > /dev/null 2>&1
  python3 - <<PY
import sqlite3, glob
dbs = glob.glob("gpurun_out/pmc_g$i/*.db") + glob.glob("gpurun_out/pmc_g$i/*/*.db")
cur = sqlite3.connect(dbs[0]).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
t = [x for x in tabs if 'counters_collection' in x or x == 'pmc_events' or 'pmc' in x.lower()]
try:
    rows = cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%gemm%' group by kernel_name, counter_name").fetchall()
    for r in rows: print(f"{r[0][:40]:40s} {r[1]:42s} {r[2]:16.1f} n={r[3]}")
except Exception as e:
    print("tables:", tabs, e)
PY
  rm -rf gpurun_out/pmc_g$i
done
