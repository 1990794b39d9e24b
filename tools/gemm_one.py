"""Run one GEMM shape a few times with a forced tile (for rocprofv3 --pmc passes): gemm_one.py M N K tile [reps] [epi] [ln]
epi 0: bias -> fp16, 1: bias + GELU -> fp16, 2: x += a w^T + b in fp32 (in place); ln = 1: the folded-LayerNorm form of the launch
(epi 2 also writes fp16(x) and the row sums; epi 0 / 1 consume (mean, rstd) and the s fragments) - what the encoders launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protosam_amd import ops
M, N, K, tile = (int(v) for v in sys.argv[1:5])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
epi = int(sys.argv[6]) if len(sys.argv) > 6 else 0
ln = int(sys.argv[7]) if len(sys.argv) > 7 else 0
dev = torch.device("cuda:0")
a = torch.randn(M, K, device=dev).half()
w = (torch.randn(N, K, device=dev) * 0.05).half()
b = torch.randn(N, device=dev)
out = torch.empty(M, N, device=dev, dtype=torch.float32 if epi == 2 else torch.float16)
if epi == 2:
    out.normal_()
kw = {}
if ln and epi == 2:
    kw = dict(out16=torch.empty(M, N, device=dev, dtype=torch.float16), stats=torch.empty(M, N // 64, 2, device=dev))
elif ln:
    st = torch.randn(M, K // 64, 2, device=dev).abs() + 1.0
    st[..., 1] = st[..., 1] * 64 + 100.0
    wf, s_ext, t_ = ops.fold_layernorm(w.float(), b, torch.ones(K, device=dev), torch.zeros(K, device=dev))
    w, b = wf, t_
    kw = dict(ln_mr=ops.ln_finalize(st, M, K, 1e-6), ln_s=s_ext)
ops.gemm_set_tile(tile)
for _ in range(reps):
    if epi == 2:
        ops.gemm(a, w, b, out=out, epilogue=ops.EPI_F32, resid=out, **kw)
    else:
        ops.gemm(a, w, b, out=out, epilogue=(ops.EPI_F16, ops.EPI_GELU_F16)[epi], **kw)
torch.cuda.synchronize()
print("done", M, N, K, tile, epi, ln)
