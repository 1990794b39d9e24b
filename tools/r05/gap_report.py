"""Where the GPU idles in a steady-state window of a rocprofv3 rocpd database (--kernel-trace): idle gaps of the union of all kernels,
grouped by the pair (kernel before, kernel after).   python3 tools/r05/gap_report.py <results.db> [frac=0.25] [out.txt]"""
import sqlite3, sys
db = sys.argv[1]; frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
rows = sqlite3.connect(db).cursor().execute("select name, start, end from kernels order by start").fetchall()
t0, t1 = rows[0][1], max(r[2] for r in rows)
cut = t1 - frac * (t1 - t0)
sel = [r for r in rows if r[1] >= cut]
span = max(r[2] for r in sel) - sel[0][1]
short = lambda n: n.replace("void ", "").split("(")[0][:48]   # noqa: E731
cur_e, cur_n = sel[0][2], sel[0][0]
pairs, total = {}, 0
for n, s, e in sel[1:]:
    if s > cur_e:
        k = (short(cur_n), short(n)); v = pairs.setdefault(k, [0, 0]); v[0] += 1; v[1] += s - cur_e; total += s - cur_e
    if e > cur_e:
        cur_e, cur_n = e, n
lines = [f"window {span / 1e6:.1f} ms, {len(sel)} dispatches, idle {total / 1e6:.2f} ms = {total / span:.3f} of it; by (kernel before -> kernel after):"]
for k, v in sorted(pairs.items(), key=lambda kv: -kv[1][1])[:45]:
    lines.append(f"  {v[1] / 1e3:9.1f} us in {v[0]:5d} gaps (avg {v[1] / v[0] / 1e3:6.1f})  {k[0]}  ->  {k[1]}")
txt = "\n".join(lines)
print(txt)
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write(txt + "\n")
