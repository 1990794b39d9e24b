#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_exp29; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_core_gpu.py -x -q -m gpu -k "attention or global" 2>&1 | tail -2
for f in pair0 pair1 pair0 pair1; do
ABL=$f PSAM_GEMM_ASM_CO=build/gattn/$f.co timeout 120 python tools/gattn_ablate.py 2>&1 | grep -v amdgpu.ids | tee -a $O/pair.txt
done
for f in pair0 pair1; do
NCALLS=10 PSAM_GEMM_ASM_CO=build/gattn/$f.co timeout 240 rocprofv3 --pmc FETCH_SIZE -d $O/pmc -o w -- python3 tools/gattn_ablate.py > /dev/null 2>&1
python3 - <<PY | tee -a $O/pair.txt
import sqlite3, glob
db = (glob.glob("$O/pmc/*.db") + glob.glob("$O/pmc/*/*.db"))[0]
cur = sqlite3.connect(db).cursor()
for r in cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like 'psam_gattn%' group by kernel_name, counter_name"):
    print(f"$f {r[0][:28]:28s} {r[1]:12s} {r[2] * 1.024e-3:10.1f} MB raw per launch n={r[3]}")
PY
rm -rf $O/pmc
done
