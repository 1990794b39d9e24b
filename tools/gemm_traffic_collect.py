"""Summarises the per-shape rocprofv3 --pmc passes of tools/gemm_traffic_by_shape.sh.
Counters are KiB per dispatch (L2 <-> fabric requests: Infinity-Cache hits are included, i.e. an upper bound of HBM traffic). On gfx950
FETCH_SIZE reports half the bytes of wide (16 B/lane) coalesced streaming reads (/opt/skills/guides/MI355X_MICROARCH.md, HBM section),
which is how the GEMM reads its operands (LDS-DMA) and its residual rows: read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE as is."""
import glob, hashlib, json, os, sqlite3, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCES = ["protosam_amd/csrc/gemm.hip", "protosam_amd/csrc/gemm_asm_gen.py", "protosam_amd/csrc/gemm_asm2_gen.py", "protosam_amd/csrc/asm_common.py"]


def git_blob_hash(path):
    data = open(os.path.join(ROOT, path), "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def source_hashes():
    return {p: git_blob_hash(p) for p in SOURCES}


def counter(dirname, name):
    dbs = glob.glob(os.path.join(dirname, "*.db")) + glob.glob(os.path.join(dirname, "*", "*.db"))
    if not dbs:
        return None
    cur = sqlite3.connect(dbs[0]).cursor()
    rows = cur.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name = ? and "
                       "(kernel_name like '%gemm%') group by kernel_name", (name,)).fetchall()
    return {r[0]: (r[1], r[2]) for r in rows}


def main(outdir, out_json, tile):
    shapes = {}
    for d in sorted(glob.glob(os.path.join(outdir, "*_FETCH_SIZE"))):
        key = os.path.basename(d)[:-len("_FETCH_SIZE")]
        M, N, K, epi = (int(v) for v in key.split("x"))
        f, w = counter(d, "FETCH_SIZE"), counter(d.replace("_FETCH_SIZE", "_WRITE_SIZE"), "WRITE_SIZE")
        if not f or not w:
            shapes[key] = {"error": "no counters collected"}
            continue
        kern = max(f, key=lambda k: f[k][0] * f[k][1])      # the kernel that moves the bytes (a split may add small ones)
        read_b = sum(2 * v[1] * 1024 * v[0] for v in f.values()) / max(f[kern][0], 1)
        write_b = sum(v[1] * 1024 * v[0] for v in w.values()) / max(w.get(kern, (1, 0))[0], 1)
        esz = 4 if epi == 2 else 2
        alg_read = 2 * (M * K + N * K) + (4 * M * N if epi == 2 else 0)
        alg_write = esz * M * N
        if epi == 2 and kern.endswith("_ln"):       # folded-LayerNorm producer: also the fp16 copy of x and the row sums
            alg_write += 2 * M * N + M * (N // 64) * 8
        shapes[key] = {"M": M, "N": N, "K": K, "epilogue": epi, "kernel": kern, "launches": f[kern][0],
                       "read_bytes_per_launch": int(read_b), "write_bytes_per_launch": int(write_b),
                       "algorithmic_read_bytes": alg_read, "algorithmic_write_bytes": alg_write,
                       "read_ratio": round(read_b / alg_read, 3), "write_ratio": round(write_b / alg_write, 3),
                       "total_ratio": round((read_b + write_b) / (alg_read + alg_write), 3)}
    doc = {"note": __doc__.strip(), "tile": int(tile), "sources": source_hashes(), "shapes": shapes}
    json.dump(doc, open(out_json, "w"), indent=1)
    for k, v in shapes.items():
        print(k, {a: v[a] for a in ("read_ratio", "write_ratio", "total_ratio", "kernel") if a in v})


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else 0)
