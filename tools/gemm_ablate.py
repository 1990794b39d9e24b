import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protosam_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, n=10, w=2):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
M, N, K = (int(v) for v in os.environ.get("SHAPE", "32768,3840,1280").split(","))
a = torch.randn(M, K, device=dev).half(); w = (torch.randn(N, K, device=dev) * 0.05).half()
out = torch.empty(M, N, device=dev, dtype=torch.float16)
ops.gemm_set_tile(int(os.environ.get("TILE", "1")))
t = timeit(lambda: ops.gemm(a, w, None, out=out, epilogue=ops.EPI_F16))
print(f"dbg={os.environ.get('PSAM_GEMM_DBG','0'):>2} tile={os.environ.get('TILE','1')} {M}x{N}x{K}: {t*1e6:8.1f} us {2*M*N*K/t/1e12:7.1f} TF/s-equivalent")
