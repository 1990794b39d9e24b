#!/bin/bash
# SQ / traffic counters of the round-5 global-attention kernels (fused rel-pos, no-bias, and the _rel form + psam_relpos they replace),
# workload tools/r05/attn_bench.py; each counter set in its own rocprofv3 --pmc pass (program directly after `--`).
# -> gpurun_out/r05_gattn_pmc_sq.txt (copy to profiles/)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_attn; rm -rf $O; mkdir -p $O
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set -d $O/p$i -o a -- python3 tools/r05/attn_bench.py > /dev/null 2>&1
done
python3 - <<'PY' > gpurun_out/r05_gattn_pmc_sq.txt
import sqlite3, glob
print("# rocprofv3 --pmc passes (one counter set per run) over tools/r05/attn_bench.py, round 5: averages per launch over ALL launches of a kernel in that")
print("# script (no-bias kernel: 16 x 1297, 1 / 2 x 1297, 1 / 4 x 5330 tokens, 12 heads; rel-pos kernels: ViT-H 16 slices and one, ViT-B 16 slices).")
print("# MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SQ_WAVE_CYCLES ... per-SIMD normalisation as in profiles/r04_gattn_pmc_sq.txt); FETCH / WRITE_SIZE in KiB (raw).")
rows = {}
for db in sorted(glob.glob("gpurun_out/pmc_attn/p*/*.db") + glob.glob("gpurun_out/pmc_attn/p*/*/*.db")):
    cur = sqlite3.connect(db).cursor()
    for r in cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%attn%' or kernel_name like '%relpos%' group by kernel_name, counter_name"):
        rows.setdefault(r[0][:44], {})[r[1]] = (r[2], r[3])
for k in sorted(rows):
    for c in sorted(rows[k]):
        v, n = rows[k][c]
        print(f"{k:44s} {c:28s} {v:16.0f} n={n}")
    d = rows[k]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "SQ_BUSY_CYCLES" in d:
        print(f"{k:44s} {'=> MFMA busy / (4 x SQ_BUSY_CYCLES / SEs...)':28s} see r04 file for the normalisation; ratio MFMA_BUSY / WAVE_CYCLES = {d['SQ_VALU_MFMA_BUSY_CYCLES'][0] / max(d['SQ_WAVE_CYCLES'][0], 1):.3f}")
PY
rm -rf $O
cat gpurun_out/r05_gattn_pmc_sq.txt
