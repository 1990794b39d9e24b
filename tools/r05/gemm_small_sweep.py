"""Round 5: which kernel family wins on the SMALL / ragged shapes that still run on the HIP 128x128 kernel (tile 1): one slice of
DINOv2-B/14 (M = 1297), two (2594), one 1022^2 slice (5330), one SAM ViT-B image (4096); epilogues as the encoders launch them.
us per call (median of 3 x 20 launches), tiles 0 (auto) / 1 / 15 / 16 / 17 (forced; ineligible shapes fall back) + torch.matmul.
  python tools/r05/gemm_small_sweep.py [tiles=0,1,15,16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from protosam_amd import ops
dev = torch.device("cuda:0")
tiles = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,1,12,15,16").split(",")]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(3):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(ts)[1]


SHAPES = []
MS = [int(v) for v in os.environ["MS"].split(",")] if os.environ.get("MS") else (1297, 2594, 4096, 5330, 8192)   # MS=20752: sixteen DINOv2 slices
for M in MS:
    SHAPES += [(M, 2304, 768, 0), (M, 768, 768, 2), (M, 3072, 768, 1), (M, 768, 3072, 2)]
if not os.environ.get("MS"):
    SHAPES += [(4096, 256, 768, 2), (4096, 256, 2304, 2), (1297, 3072, 1024, 0), (1297, 1024, 1024, 2), (1297, 4096, 1024, 1), (1297, 1024, 4096, 2),
               (4096, 3840, 1280, 0), (4096, 1280, 1280, 2), (4096, 5120, 1280, 1), (4096, 1280, 5120, 2)]
for (M, N, K, epi) in SHAPES:
    a = torch.randn(M, K, device=dev).half()
    w = (torch.randn(N, K, device=dev) * 0.05).half()
    bias, gamma = torch.randn(N, device=dev), torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if epi == 2 else torch.float16)
    o16 = torch.empty(M, N, device=dev, dtype=torch.float16)
    res = []
    for tl in tiles:
        ops.gemm_set_tile(tl)
        if epi == 2:
            t = timeit(lambda: ops.gemm(a, w, bias, out=out, epilogue=ops.EPI_F32, resid=out, gamma=gamma))
        else:
            t = timeit(lambda: ops.gemm(a, w, bias, out=out, epilogue=epi))
        res.append(f"t{tl} {t:6.1f}")
    ops.gemm_set_tile(0)
    t = timeit(lambda: torch.matmul(a, w.t(), out=o16))
    res.append(f"lib {t:6.1f}")
    print(f"{M:5d} x {N:4d} x {K:4d} epi {epi}: " + "  ".join(res) + f"   ({2.0 * M * N * K / 1e9:.1f} GF)", flush=True)
