"""Run one GEMM shape a few times with a forced tile (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protosam_amd import ops
M, N, K, tile = (int(v) for v in sys.argv[1:5])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
dev = torch.device("cuda:0")
a = torch.randn(M, K, device=dev).half()
w = (torch.randn(N, K, device=dev) * 0.05).half()
b = torch.randn(N, device=dev)
out = torch.empty(M, N, device=dev, dtype=torch.float16)
ops.gemm_set_tile(tile)
for _ in range(reps):
    ops.gemm(a, w, b, out=out, epilogue=ops.EPI_F16)
torch.cuda.synchronize()
print("done", M, N, K, tile)
