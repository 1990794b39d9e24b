"""Builds profiles/<tag>_gemm_pmc_traffic.json from two rocprofv3 --pmc databases (FETCH_SIZE pass, WRITE_SIZE pass) of the
same bench command.  usage: pmc_traffic.py <fetch.db> <write.db> <out.json> "<command>"
Counters are KiB per dispatch; on gfx950 FETCH_SIZE reports half the bytes of wide (16 B/lane) coalesced streaming reads
(MI355X_MICROARCH.md, HBM section) - what the GEMM's global_load_lds DMA issues - so read bytes = 2 * FETCH_SIZE * 1024."""
import json
import sqlite3
import sys


def per_kernel(db, counter):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name = ? "
                       "and kernel_name like '%gemm%' group by kernel_name", (counter,)).fetchall()
    return {r[0]: (r[1], r[2]) for r in rows}


def main(fetch_db, write_db, out, command):
    f, w = per_kernel(fetch_db, "FETCH_SIZE"), per_kernel(write_db, "WRITE_SIZE")
    kernels, tot_b, tot_n = {}, 0.0, 0
    for k in sorted(f):
        n, fk = f[k]
        wk = w.get(k, (0, 0.0))[1]
        b = 2 * fk * 1024 + wk * 1024
        kernels[k] = {"launches": n, "fetch_kib_avg": round(fk, 1), "write_kib_avg": round(wk, 1), "bytes_per_launch": int(b)}
        tot_b += b * n
        tot_n += n
    doc = {"command": command,
           "note": "Counters are KiB per dispatch. Per /opt/skills/guides/MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 "
                   "reports half of the bytes of wide (16 B/lane) coalesced streaming reads, which is what the GEMM's "
                   "global_load_lds DMA issues, so read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is taken as is. The counters "
                   "are L2<->fabric requests (Infinity-Cache hits included), i.e. an upper bound of true HBM traffic.",
           "kernels": kernels, "all_gemm_launches": {"launches": tot_n, "bytes_per_launch_avg": int(tot_b / max(tot_n, 1))}}
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(doc["all_gemm_launches"]))


if __name__ == "__main__":
    main(*sys.argv[1:5])
