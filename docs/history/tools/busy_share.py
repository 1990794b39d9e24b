"""GPU-busy share of a steady-state window of a rocprofv3 rocpd database (`--kernel-trace`): the last `frac` of the traced span.
  python3 tools/busy_share.py <results.db> [frac=0.25] [out.txt]"""
import sqlite3, sys
db = sys.argv[1]; frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
rows = sqlite3.connect(db).cursor().execute("select name, start, end from kernels order by start").fetchall()
t0, t1 = rows[0][1], max(r[2] for r in rows)
cut = t1 - frac * (t1 - t0)
sel = [r for r in rows if r[1] >= cut]
span = max(r[2] for r in sel) - sel[0][1]
iv = sorted((r[1], r[2]) for r in sel)
union, cur_s, cur_e = 0, iv[0][0], iv[0][1]
gaps = []
for s, e in iv[1:]:
    if s > cur_e:
        union += cur_e - cur_s; gaps.append(s - cur_e); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
by = {}
for n, s, e in sel:
    k = n[:70]; by.setdefault(k, [0, 0]); by[k][0] += 1; by[k][1] += e - s
lines = [f"steady window {span / 1e6:.1f} ms: {len(sel)} dispatches, kernel time {sum(e - s for _, s, e in sel) / 1e6:.1f} ms, "
         f"GPU busy (union) {union / 1e6:.1f} ms = {union / span:.3f}; idle gaps: n {len(gaps)}, sum {sum(gaps) / 1e6:.1f} ms, "
         f"median {sorted(gaps)[len(gaps) // 2] / 1e3:.1f} us, >20us: {sum(1 for g in gaps if g > 20000)} ({sum(g for g in gaps if g > 20000) / 1e6:.1f} ms)"]
for k, v in sorted(by.items(), key=lambda kv: -kv[1][1])[:30]:
    lines.append(f"  {v[1] / 1e6:8.2f} ms {v[0]:6d}  {k}")
txt = "\n".join(lines)
print(txt)
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write(txt + "\n")
