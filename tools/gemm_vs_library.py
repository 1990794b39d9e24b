"""Our GEMM launches (WITH their epilogues, as the encoders issue them: folded-LayerNorm forms where the 16-slice path uses them)
against the library GEMM torch dispatches to (hipBLASLt / rocBLAS: plain fp16 GEMM, no bias / activation / residual) on the SAME
tensors in one process, interleaved.   python tools/gemm_vs_library.py > profiles/rNN_gemm_vs_library.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protosam_amd import ops
dev = torch.device("cuda:0")


def timed(fn, n=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


print("shape (M x N x K), epilogue | ours TFLOP/s (plain form / folded-LayerNorm form) | torch.matmul TFLOP/s (no epilogue) | ours / library")
for (M, N, K, epi, what) in [(65536, 3840, 1280, 0, "SAM-H qkv"), (65536, 1280, 1280, 2, "SAM-H proj"), (65536, 5120, 1280, 1, "SAM-H fc1 + GELU"),
                             (65536, 1280, 5120, 2, "SAM-H fc2"), (65536, 2304, 768, 0, "SAM-B qkv"), (65536, 768, 768, 2, "SAM-B proj"),
                             (65536, 3072, 768, 1, "SAM-B fc1 + GELU"), (65536, 768, 3072, 2, "SAM-B fc2"),
                             (20752, 2304, 768, 0, "DINOv2-B qkv (16 slices)"), (20752, 768, 3072, 2, "DINOv2-B fc2")]:
    a = torch.randn(M, K, device=dev).half()
    w = (torch.randn(N, K, device=dev) * 0.05).half()
    b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if epi == 2 else torch.float16)
    o16 = torch.empty(M, N, device=dev, dtype=torch.float16)
    if epi == 2:
        out.normal_()
        x16 = torch.empty(M, N, device=dev, dtype=torch.float16); stats = torch.empty(M, N // 64, 2, device=dev)
        plain = lambda: ops.gemm(a, w, b, out=out, epilogue=ops.EPI_F32, resid=out)
        ln = lambda: ops.gemm(a, w, b, out=out, epilogue=ops.EPI_F32, resid=out, out16=x16, stats=stats)
    else:
        e = (ops.EPI_F16, ops.EPI_GELU_F16)[epi]
        st = torch.randn(M, K // 64, 2, device=dev).abs() + 1.0
        st[..., 1] = st[..., 1] * 64 + 100.0
        wf, s_ext, t_ = ops.fold_layernorm(w.float(), b, torch.ones(K, device=dev), torch.zeros(K, device=dev))
        mr = ops.ln_finalize(st, M, K, 1e-6)
        plain = lambda: ops.gemm(a, w, b, out=out, epilogue=e)
        ln = lambda: ops.gemm(a, wf, t_, out=out, epilogue=e, ln_mr=mr, ln_s=s_ext)
    wt = w.t()
    lib = lambda: torch.matmul(a, wt, out=o16)
    fl = 2.0 * M * N * K
    r = {"plain": [], "ln": [], "lib": []}
    for rep in range(3):
        r["plain"].append(fl / timed(plain) / 1e12)
        r["lib"].append(fl / timed(lib) / 1e12)
        r["ln"].append(fl / timed(ln) / 1e12)
    med = lambda v: sorted(v)[len(v) // 2]
    print(f"{M} x {N} x {K}, epilogue {epi} ({what}) | {med(r['plain']):6.0f} / {med(r['ln']):6.0f} | {med(r['lib']):6.0f} | "
          f"{med(r['plain']) / med(r['lib']):.2f} / {med(r['ln']) / med(r['lib']):.2f}", flush=True)
