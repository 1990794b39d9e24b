"""Per-slice shapes of the fp32-residual GEMM (EPI 2, in place): automatic choice (split-K where the rule applies) against the
forced single-pass kernels, in-process. usage: gemm_splitk_bench.py [M N K]..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protosam_amd import ops
dev = torch.device("cuda:0")
shapes = [(4096, 1280, 5120), (4096, 1280, 1280), (1297, 768, 3072), (1297, 768, 768), (4096, 768, 3072), (4096, 1024, 4096)]


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for M, N, K in shapes:
    a = torch.randn(M, K, device=dev).half()
    w = (torch.randn(N, K, device=dev) * 0.05).half()
    b = torch.randn(N, device=dev)
    x = torch.randn(M, N, device=dev)
    res = {}
    for tile in (0, 1, 11):
        ops.gemm_set_tile(tile)
        res[tile] = timed(lambda: ops.gemm(a, w, b, out=x, epilogue=ops.EPI_F32, resid=x))
    ops.gemm_set_tile(0)
    fl = 2.0 * M * N * K
    print(f"{M}x{N}x{K}: " + "  ".join(f"tile {t}: {us:.1f} us ({fl / us / 1e6:.0f} TF)" for t, us in res.items()))

# fp16-output shapes of one slice (column split of a just-over-one-round tile count)
for M, N, K, epi in ((4096, 5120, 1280, ops.EPI_GELU_F16), (4096, 3840, 1280, ops.EPI_F16), (4096, 4096, 1024, ops.EPI_GELU_F16),
                     (4096, 3072, 768, ops.EPI_GELU_F16)):
    a = torch.randn(M, K, device=dev).half()
    w = (torch.randn(N, K, device=dev) * 0.05).half()
    b = torch.randn(N, device=dev)
    o = torch.empty(M, N, device=dev, dtype=torch.float16)
    res = {}
    for tile in (0, 1, 11):
        ops.gemm_set_tile(tile)
        res[tile] = timed(lambda: ops.gemm(a, w, b, out=o, epilogue=epi))
    ops.gemm_set_tile(0)
    ref = o.clone()
    ops.gemm_set_tile(1)
    ops.gemm(a, w, b, out=o, epilogue=epi)
    ops.gemm_set_tile(0)
    ops.gemm(a, w, b, out=ref, epilogue=epi)
    fl = 2.0 * M * N * K
    print(f"{M}x{N}x{K} epi {epi}: " + "  ".join(f"tile {t}: {us:.1f} us ({fl / us / 1e6:.0f} TF)" for t, us in res.items()),
          "auto == tile 1:", bool(torch.equal(o, ref)))
