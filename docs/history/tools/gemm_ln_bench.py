"""The folded-LayerNorm producer GEMM (x += a w^T + b, plus fp16(x) and the row sums) against the plain one, TFLOP/s.
   python tools/gemm_ln_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protosam_amd import ops
dev = torch.device("cuda:0")


def timeit(fn, n=8, w=2):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


for (M, N, K) in [(65536, 1280, 1280), (65536, 1280, 5120)]:
    a = torch.randn(M, K, device=dev).half(); w = (torch.randn(N, K, device=dev) * 0.05).half()
    bias = torch.randn(N, device=dev); x = torch.randn(M, N, device=dev)
    x16 = torch.empty(M, N, device=dev, dtype=torch.float16); stats = torch.empty(M, N // 64, 2, device=dev)
    res = []
    for rep in range(2):
        t = timeit(lambda: ops.gemm(a, w, bias, out=x, epilogue=ops.EPI_F32, resid=x))
        res.append(f"plain={2*M*N*K/t/1e12:6.0f}")
        t = timeit(lambda: ops.gemm(a, w, bias, out=x, epilogue=ops.EPI_F32, resid=x, out16=x16, stats=stats))
        res.append(f"ln={2*M*N*K/t/1e12:6.0f}")
    print(f"{os.environ.get('PSAM_LIB_PATH', 'shipped')[-8:]} {M}x{N}x{K}: " + " ".join(res), flush=True)
