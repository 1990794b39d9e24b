import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from protosam_amd import ops
dev = torch.device("cuda:0")
M, N, K = 65536, 1280, 1280
a = torch.randn(M, K, device=dev).half(); w = (torch.randn(N, K, device=dev) * 0.05).half()
bias = torch.randn(N, device=dev)
out = torch.randn(M, N, device=dev)
other = torch.randn(M, N, device=dev)
ops.gemm_set_tile(15); ops.gemm_asm_variant(1)
for name, kw in (("inplace resid", dict(resid=out)), ("no resid", dict(resid=None)), ("resid other buffer", dict(resid=other))):
    print(name, file=sys.stderr, flush=True)
    for _ in range(3):
        ops.gemm(a, w, bias, out=out, epilogue=ops.EPI_F32, **kw)
    torch.cuda.synchronize()
o16 = torch.empty(M, N, device=dev, dtype=torch.float16)
print("fp16 out, same shape", file=sys.stderr, flush=True)
for _ in range(3):
    ops.gemm(a, w, bias, out=o16, epilogue=ops.EPI_F16)
torch.cuda.synchronize()
