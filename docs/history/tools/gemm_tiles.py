"""Within-run A/B of the GEMM tile variants on a list of shapes (random data), plus torch.matmul (hipBLASLt) beside them.
  python tools/gemm_tiles.py [tiles=1,5,7] [shapes=MxNxK;...]"""
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


tiles = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1,5,7").split(",")]
shapes = sys.argv[2] if len(sys.argv) > 2 else "8192x8192x8192;4096x4096x4096;32768x3840x1280;32768x5120x1280;32768x1280x5120;32768x1280x1280;4096x3840x1280;10376x2304x768"
for sh in shapes.split(";"):
    M, N, K = (int(v) for v in sh.split("x"))
    a = torch.randn(M, K, device=dev).half(); w = (torch.randn(N, K, device=dev) * 0.05).half()
    f32 = len(sys.argv) > 3 and sys.argv[3] == "f32"     # the `x += gamma * (a @ w^T + b)` epilogue, in place
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if f32 else torch.float16)
    bias = torch.randn(N, device=dev); gamma = torch.randn(N, device=dev)
    res = []
    for rep in range(2):
        for tl in tiles:
            if tl in (3, 5, 7) and N % 256:
                continue
            ops.gemm_set_tile(tl)
            if f32:
                t = timeit(lambda: ops.gemm(a, w, bias, out=out, epilogue=ops.EPI_F32, resid=out, gamma=gamma))
            else:
                t = timeit(lambda: ops.gemm(a, w, None, out=out, epilogue=ops.EPI_F16))
            res.append(f"t{tl}={2*M*N*K/t/1e12:6.0f}")
        o16 = out if not f32 else torch.empty(M, N, device=dev, dtype=torch.float16)
        t = timeit(lambda: torch.matmul(a, w.t(), out=o16))
        res.append(f"blaslt={2*M*N*K/t/1e12:6.0f}")
    print(f"{sh:>18}: " + " ".join(res), flush=True)
ops.gemm_set_tile(0)
