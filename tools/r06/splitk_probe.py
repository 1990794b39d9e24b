"""One SAM ViT-H image through mlp.lin2 (4096 x 1280 x 5120) + the LayerNorm behind it: the half-tile kernel + LayerNorm pass the one-slice
path runs today against psam_gemm_f16_splitk_ln (K ranges of the assembly tile + one reduce / LayerNorm pass). Timed inside replayed graphs
(ten calls each) so that launch overhead of the host is not in the figure."""
import sys
import torch
sys.path.insert(0, ".")
from protosam_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)


def timeit(fn, n=20):
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


for (M, N, K) in ((4096, 1280, 5120), (4096, 1024, 4096), (8192, 1280, 5120)):
    a = (torch.randn((M, K), generator=g)).half().to(dev)
    w = (torch.randn((N, K), generator=g) * 0.02).half().to(dev)
    b = torch.randn(N, generator=g).to(dev)
    x = torch.randn((M, N), generator=g).to(dev)
    lnw, lnb = torch.ones(N, device=dev), torch.zeros(N, device=dev)
    ln16 = torch.empty((M, N), dtype=torch.float16, device=dev)

    def old():
        ops.gemm(a, w, b, out=x, epilogue=ops.EPI_F32, resid=x)
        ops.layernorm(x, lnw, lnb, 1e-6, out=ln16)
    old()
    torch.cuda.synchronize()
    g0 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g0):
        for _ in range(10):
            old()
    t_old = timeit(g0.replay) / 10
    ks = ops.gemm_splitk_ranges(M, N, K)
    if ks < 2:
        print(f"{M}x{N}x{K}: one launch + LayerNorm pass {t_old:.1f} us; split-K declined")
        continue
    ws = torch.empty((ks, ops.splitk_rows(M), N), dtype=torch.float32, device=dev)

    def new():
        ops.gemm_splitk_ln(a, w, b, x, ks, ws, lnw, lnb, 1e-6, out16=ln16)
    new()
    torch.cuda.synchronize()
    g1 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1):
        for _ in range(10):
            new()
    t_new = timeit(g1.replay) / 10
    print(f"{M}x{N}x{K}: half-tile GEMM + LayerNorm pass {t_old:.1f} us; {ks} K ranges + reduce / LayerNorm pass {t_new:.1f} us")
