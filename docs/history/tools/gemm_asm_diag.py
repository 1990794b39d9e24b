"""Diagnostics of the assembly GEMM on one small shape: which elements get written, block-wise error maps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protosam_amd import ops
dev = torch.device("cuda:0")
M, N, K = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "256x256x128").split("x"))
torch.manual_seed(0)
a = torch.randn(M, K, device=dev).half()
w = (torch.randn(N, K, device=dev) * 0.05).half()
ref = a.float() @ w.float().t()


def blockmap(t, bs=32):
    m = t.view(M // bs, bs, N // bs, bs).amax(dim=(1, 3))
    return m


torch.set_printoptions(linewidth=250, precision=3, sci_mode=False)
# fp32 epilogue, nothing else: out = acc
x = torch.full((M, N), 777.0, device=dev)
ops.gemm_set_tile(15)
ops.gemm(a, w, None, out=x, epilogue=ops.EPI_F32, resid=None)
torch.cuda.synchronize()
written = (x != 777.0)
print("fp32: written fraction", written.float().mean().item())
print("written per 32x32 block:\n", blockmap(written.float()))
err = (x - ref).abs()
print("err max per 32x32 block:\n", blockmap(err))
# is the output a permutation problem? check whether out matches ref at some shifted rows
if M == 256 and N == 256:
    r0 = x[:32, :32]
    best = None
    for bi in range(8):
        for bj in range(8):
            d = (r0 - ref[bi * 32:(bi + 1) * 32, bj * 32:(bj + 1) * 32]).abs().max().item()
            dt = (r0 - ref[bi * 32:(bi + 1) * 32, bj * 32:(bj + 1) * 32].t()).abs().max().item()
            if best is None or min(d, dt) < best[0]:
                best = (min(d, dt), bi, bj, d < dt)
    print("block (0,0) of out best matches ref block", best)
# K-prefix test: does out equal the product over only part of K?
for kk in range(64, K + 1, 64):
    part = a[:, :kk].float() @ w[:, :kk].float().t()
    print("vs K-prefix", kk, (x - part).abs().max().item())
for k0 in range(0, K, 64):
    part = a[:, k0:k0 + 64].float() @ w[:, k0:k0 + 64].float().t()
    print("vs K-tile", k0 // 64, "alone", (x - part).abs().max().item())
# fp16 epilogue
o = torch.full((M, N), 3.0, device=dev, dtype=torch.float16)
ops.gemm(a, w, None, out=o, epilogue=ops.EPI_F16)
torch.cuda.synchronize()
wr = (o != 3.0)
print("fp16: written fraction", wr.float().mean().item())
print("written per block:\n", blockmap(wr.float()))
print("err per block:\n", blockmap((o.float() - ref).abs()))
ops.gemm_set_tile(0)
