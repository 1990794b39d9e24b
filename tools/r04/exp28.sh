#!/bin/bash
# SQ counters of the two assembly attention kernels (window: tools/wattn_time.py; global: tools/gattn_ablate.py with the shipped build)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_exp28; mkdir -p $O
for prog in wattn_time gattn_ablate; do
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
NCALLS=20 timeout 240 rocprofv3 --pmc $set -d $O/pmc -o w -- python3 tools/$prog.py > /dev/null 2>&1
python3 - <<PY | tee -a $O/sq_$prog.txt
import sqlite3, glob
db = (glob.glob("$O/pmc/*.db") + glob.glob("$O/pmc/*/*.db"))[0]
cur = sqlite3.connect(db).cursor()
for r in cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like 'psam_%attn%' group by kernel_name, counter_name"):
    print(f"{r[0][:28]:28s} {r[1]:28s} {r[2]:16.0f} n={r[3]}")
PY
rm -rf $O/pmc
done
done
