"""Per-tile fixed cost vs per-K-step cost of a GEMM tile variant: time M x N x K for a sweep of K and fit t = a + b*K/64
per round of 256 workgroups.   python tools/gemm_ksweep.py [tile=10] [M=65536] [N=3840] [f32]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from protosam_amd import ops
dev = torch.device("cuda:0")
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 10
M = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
N = int(sys.argv[3]) if len(sys.argv) > 3 else 3840
f32 = len(sys.argv) > 4 and sys.argv[4] == "f32"


def timeit(fn, n=8, w=2):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


ops.gemm_set_tile(tile)
rounds = ((M + 255) // 256) * (N // 256) / 256.0
xs, ys = [], []
for K in (64, 128, 256, 512, 1024, 1280, 2560, 5120):
    a = torch.randn(M, K, device=dev).half(); w = (torch.randn(N, K, device=dev) * 0.05).half()
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if f32 else torch.float16)
    bias = torch.randn(N, device=dev); gamma = torch.randn(N, device=dev)
    if f32:
        t = timeit(lambda: ops.gemm(a, w, bias, out=out, epilogue=ops.EPI_F32, resid=out, gamma=gamma))
    else:
        t = timeit(lambda: ops.gemm(a, w, None, out=out, epilogue=ops.EPI_F16))
    o16 = torch.empty(M, N, device=dev, dtype=torch.float16)
    tb = timeit(lambda: torch.matmul(a, w.t(), out=o16))
    print(f"K={K:5d}: {t*1e6:8.1f} us  {2*M*N*K/t/1e12:6.0f} TF/s   per round {t*1e6/rounds:6.2f} us   blaslt {tb*1e6:8.1f} us {2*M*N*K/tb/1e12:6.0f}", flush=True)
    xs.append(K / 64); ys.append(t * 1e6 / rounds)
b, a = np.polyfit(xs[2:], ys[2:], 1)
print(f"fit (K>=256): per-round fixed a = {a:.2f} us, per K-step b = {b:.3f} us  (rounds = {rounds:.2f})")
ops.gemm_set_tile(0)
