"""Summarise a rocprofv3 rocpd sqlite database (`--kernel-trace --stats` output of ROCm 7.2) as a per-kernel table
(markdown), the form committed under profiles/."""
import sqlite3
import sys


def main(db_path, out_path=None, note=""):
    cur = sqlite3.connect(db_path).cursor()
    rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    lines = ["# rocprofv3 --kernel-trace --stats summary", "", note, "",
             f"total kernel time: {tot / 1e6:.2f} ms over {sum(r[1] for r in rows)} dispatches", "",
             "| kernel | calls | total ms | % | avg us | min us | max us |", "|---|---|---|---|---|---|---|"]
    for r in rows:
        lines.append(f"| `{r[0][:100]}` | {r[1]} | {r[2] / 1e6:.3f} | {100 * r[2] / tot:.1f} | {r[3] / 1e3:.1f} | "
                     f"{r[4] / 1e3:.1f} | {r[5] / 1e3:.1f} |")
    txt = "\n".join(lines) + "\n"
    if out_path:
        open(out_path, "w").write(txt)
    else:
        print(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None, sys.argv[3] if len(sys.argv) > 3 else "")
