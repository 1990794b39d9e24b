#!/bin/bash
# per-slice path: wall vs kernel time (GPU-busy share), and the sustained clock / power under the batched load
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_exp15; mkdir -p $O
python3 tools/per_slice_profile.py 1 16 auto 3 2>&1 | tail -1 | tee $O/wall.txt
python3 tools/per_slice_profile.py 1 16 0 3 2>&1 | tail -1 | tee -a $O/wall.txt
SMI=1 SMI_OUT=$O/smi_batch16.csv python3 tools/per_slice_profile.py 16 16 0 40 2>&1 | tail -1 | tee -a $O/wall.txt
SMI=1 SMI_OUT=$O/smi_batch1.csv python3 tools/per_slice_profile.py 1 16 auto 20 2>&1 | tail -1 | tee -a $O/wall.txt
rocm-smi -P -c > $O/smi_idle.txt 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof -o ps -- python3 tools/per_slice_profile.py 1 16 0 3 > $O/prof_run.txt 2>&1
tail -1 $O/prof_run.txt
python3 - <<'PY'
import csv, glob, os
O = "gpurun_out/r04_exp15"
f = glob.glob(O + "/prof/**/*kernel_trace.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    # the last 3 x 16 slices: take the dispatches of the last 40 % of the time span as steady state
    t0 = min(int(r["Start_Timestamp"]) for r in rows); t1 = max(int(r["End_Timestamp"]) for r in rows)
    cut = t1 - 0.25 * (t1 - t0)
    sel = [r for r in rows if int(r["Start_Timestamp"]) >= cut]
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sel)
    span = max(int(r["End_Timestamp"]) for r in sel) - min(int(r["Start_Timestamp"]) for r in sel)
    # union of busy intervals (single stream: no overlap)
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in sel)
    gaps = [b[0] - a[1] for a, b in zip(iv, iv[1:]) if b[0] > a[1]]
    by = {}
    for r in sel:
        k = r["Kernel_Name"][:60]
        by.setdefault(k, [0, 0]); by[k][0] += 1; by[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    with open(O + "/busy.txt", "w") as fo:
        print(f"steady window {span/1e6:.1f} ms: {len(sel)} dispatches, kernel time {busy/1e6:.1f} ms = {busy/span:.3f} busy; gaps: n {len(gaps)}, sum {sum(gaps)/1e6:.1f} ms, median {sorted(gaps)[len(gaps)//2]/1e3:.1f} us", file=fo)
        for k, v in sorted(by.items(), key=lambda kv: -kv[1][1])[:25]:
            print(f"  {v[1]/1e6:8.2f} ms {v[0]:6d}  {k}", file=fo)
    print(open(O + "/busy.txt").read())
rm = None
PY
rm -rf $O/prof
