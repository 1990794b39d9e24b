"""Which kernels a STEADY-STATE step launches: from a rocprofv3 --kernel-trace database of a bench.py run, the dispatches between
consecutive launches of the step's first kernel (DINOv2's `patchify_bilinear_kernel` on the query batch; the support is cached), averaged
over the last `n` steps - setup, weight packing and warm-up are outside the window. Marks the stock PyTorch / runtime kernels.

  python tools/step_kernels.py <db> [n_steps] > profiles/rNN_step_kernels.md
"""
import sqlite3
import sys


def main(db, n=5):
    cur = sqlite3.connect(db).cursor()
    marks = [r[0] for r in cur.execute("select start from kernels where name like '%patchify_bilinear_kernel%' order by start").fetchall()]
    assert len(marks) >= n + 1, len(marks)
    t0, t1 = marks[-(n + 1)], marks[-1]
    rows = cur.execute("select name, count(*), sum(end-start) from kernels where start >= ? and start < ? group by name order by 3 desc",
                       (t0, t1)).fetchall()
    tot = sum(r[2] for r in rows)
    ncall = sum(r[1] for r in rows)
    stock = lambda nm: nm.startswith("void at::") or "rocclr" in nm or "rocblas" in nm or nm.startswith("void (anonymous namespace)") or "at::native" in nm  # noqa: E731
    st = [r for r in rows if stock(r[0])]
    print(f"# kernels of a steady-state step (mean of the last {n} steps of the traced run)\n")
    print(f"{ncall / n:.0f} dispatches and {tot / n / 1e6:.2f} ms of kernel time per step; wall time between step starts {(t1 - t0) / n / 1e6:.2f} ms; "
          f"stock PyTorch / runtime kernels: {sum(r[1] for r in st) / n:.0f} dispatches, {sum(r[2] for r in st) / n / 1e6:.3f} ms "
          f"({100 * sum(r[2] for r in st) / max(tot, 1):.2f} % of the kernel time)\n")
    print("| kernel | calls / step | ms / step | % | stock |\n|---|---|---|---|---|")
    for name, c, t in rows:
        print(f"| `{name[:110]}` | {c / n:.1f} | {t / n / 1e6:.3f} | {100 * t / tot:.2f} | {'yes' if stock(name) else ''} |")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 5)
