"""Run one GEMM shape a few times with a forced tile (for rocprofv3 --pmc passes): gemm_one.py M N K tile [reps] [epi]
epi 0: bias -> fp16, 1: bias + GELU -> fp16, 2: x += a w^T + b in fp32 (in place)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protosam_amd import ops
M, N, K, tile = (int(v) for v in sys.argv[1:5])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
epi = int(sys.argv[6]) if len(sys.argv) > 6 else 0
dev = torch.device("cuda:0")
a = torch.randn(M, K, device=dev).half()
w = (torch.randn(N, K, device=dev) * 0.05).half()
b = torch.randn(N, device=dev)
out = torch.empty(M, N, device=dev, dtype=torch.float32 if epi == 2 else torch.float16)
if epi == 2:
    out.normal_()
ops.gemm_set_tile(tile)
for _ in range(reps):
    if epi == 2:
        ops.gemm(a, w, b, out=out, epilogue=ops.EPI_F32, resid=out)
    else:
        ops.gemm(a, w, b, out=out, epilogue=(ops.EPI_F16, ops.EPI_GELU_F16)[epi])
torch.cuda.synchronize()
print("done", M, N, K, tile, epi)
