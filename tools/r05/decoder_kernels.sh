#!/bin/bash
# kernels of one MaskDecoder.predict_masks_tokens call (27 prompt sets on 16 images), by (kernel, grid): count and average duration
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace -d $O/prof_dk -o t -- python3 tools/r05/decoder_bench.py > $O/r05_decoder_bench.log 2>&1
DB=$(ls $O/prof_dk/*.db $O/prof_dk/*/*.db 2>/dev/null | head -1)
python3 - $DB <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1]).cursor()
cols = [r[1] for r in c.execute("pragma table_info(kernels)").fetchall()]
gx = [k for k in cols if "grid" in k.lower()]
rows = c.execute("select name, start, end, %s from kernels order by start" % ", ".join(gx)).fetchall()
n = len(rows); rows = rows[int(n * 0.6):]          # the timed calls
by = {}
for r in rows:
    k = (r[0].replace("void ", "").split("(")[0][:44], tuple(r[3:]))
    v = by.setdefault(k, [0, 0]); v[0] += 1; v[1] += r[2] - r[1]
tot = sum(v[1] for v in by.values())
print("columns:", gx, " total %.1f ms" % (tot / 1e6))
for k, v in sorted(by.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{v[1] / tot * 100:5.1f} %  {v[0]:4d} x {v[1] / v[0] / 1e3:7.1f} us  {k[0]:44s} grid {k[1]}")
PY
rm -rf $O/prof_dk
