import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protosam_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, n=20, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for (M, N, K) in [(4096, 3840, 1280), (4096, 1280, 1280), (4096, 5120, 1280), (4096, 1280, 5120), (8192, 5120, 1280), (16384, 1280, 5120), (16384, 3840, 1280), (10376, 2304, 768), (8192, 8192, 8192)]:
    a = torch.randn(M, K, device=dev).half(); w = (torch.randn(N, K, device=dev) * 0.05).half()
    out = torch.empty(M, N, device=dev, dtype=torch.float16)
    ref = (a[:64].float() @ w.float().t())
    for tile in (1, 2, 3):
        if tile == 3 and N % 256: continue
        ops.gemm_set_tile(tile)
        t = timeit(lambda: ops.gemm(a, w, None, out=out, epilogue=ops.EPI_F16))
        err = (out[:64].float() - ref).abs().max().item()
        print(f"map={os.environ.get('PSAM_GEMM_MAP','0')} {M}x{N}x{K} tile{tile}: {t*1e6:8.1f} us {2*M*N*K/t/1e12:7.1f} TF/s err {err:.3f}", flush=True)
