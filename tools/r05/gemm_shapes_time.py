"""Round 5: the four SAM ViT-H block GEMMs at 16 slices as the pipeline launches them (folded-LayerNorm forms), us per launch and
TFLOP/s, for A/B of code objects (PSAM_GEMM_ASM_CO=build/<variant>.co).   python tools/r05/gemm_shapes_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from protosam_amd import ops
dev = torch.device("cuda:0")
M, D = 65536, 1280
torch.manual_seed(0)
x = torch.randn(M, D, device=dev)
x16 = torch.empty(M, D, device=dev, dtype=torch.float16)
stats = torch.empty(M, D // 64, 2, device=dev)
mr = ops.ln_mr_buffer(M, dev)
att = torch.randn(M, D, device=dev).half()
hid = torch.randn(M, 4 * D, device=dev).half()
qkv = torch.empty(M, 3 * D, device=dev, dtype=torch.float16)
hid_o = torch.empty(M, 4 * D, device=dev, dtype=torch.float16)
lnw, lnb = torch.randn(D, device=dev), torch.randn(D, device=dev)
Wq, bq = torch.randn(3 * D, D, device=dev) * 0.05, torch.randn(3 * D, device=dev)
W1, b1 = torch.randn(4 * D, D, device=dev) * 0.05, torch.randn(4 * D, device=dev)
qw, qs, qt = ops.fold_layernorm(Wq, bq, lnw, lnb)
fw, fs, ft = ops.fold_layernorm(W1, b1, lnw, lnb)
pw, pb = (torch.randn(D, D, device=dev) * 0.05).half(), torch.randn(D, device=dev)
w2, b2 = (torch.randn(D, 4 * D, device=dev) * 0.05).half(), torch.randn(D, device=dev)
ops.gemm(att, pw, pb, out=x, epilogue=ops.EPI_F32, resid=x, out16=x16, stats=stats)
ops.ln_finalize(stats, M, D, 1e-6, mr=mr)
cases = [("qkv  (f16_ln)", lambda: ops.gemm(x16, qw, qt, out=qkv, epilogue=ops.EPI_F16, ln_mr=mr, ln_s=qs), 2.0 * M * 3 * D * D),
         ("proj (f32_ln)", lambda: ops.gemm(att, pw, pb, out=x, epilogue=ops.EPI_F32, resid=x, out16=x16, stats=stats), 2.0 * M * D * D),
         ("fc1  (gelu_ln)", lambda: ops.gemm(x16, fw, ft, out=hid_o, epilogue=ops.EPI_GELU_F16, ln_mr=mr, ln_s=fs), 2.0 * M * 4 * D * D),
         ("fc2  (f32_ln)", lambda: ops.gemm(hid, w2, b2, out=x, epilogue=ops.EPI_F32, resid=x, out16=x16, stats=stats), 2.0 * M * 4 * D * D)]
res = {n: [] for n, _, _ in cases}
for rep in range(3):
    for n, fn, fl in cases:
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[n].append(e0.elapsed_time(e1) / 20 * 1e3)
tot = 0.0
out = []
for n, fn, fl in cases:
    t = sorted(res[n])[1]
    tot += t
    out.append(f"{n} {t:6.1f} us {fl / t / 1e6:5.0f} TF/s")
print(f"{os.environ.get('PSAM_GEMM_ASM_CO', 'shipped'):>22}: " + " | ".join(out) + f" | block sum {tot:.0f} us", flush=True)
