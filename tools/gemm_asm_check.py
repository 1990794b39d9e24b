"""Correctness + speed of the assembly GEMM (tile 15) against the HIP kernels (tile 11 / 1) and an fp32 product.
  python tools/gemm_asm_check.py [quick|full] [bench]
Every case prints as it finishes (a hang shows where)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protosam_amd import ops
dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "quick"
bench = "bench" in sys.argv[2:]
TILE = 16 if "t16" in sys.argv[2:] else 15


def run(tile, a, w, bias, epi, resid, gamma):
    ops.gemm_set_tile(tile)
    try:
        if epi == 2:
            x = resid.clone() if resid is not None else None
            return ops.gemm(a, w, bias, out=x, epilogue=ops.EPI_F32, resid=x, gamma=gamma)
        return ops.gemm(a, w, bias, epilogue=(ops.EPI_F16, ops.EPI_GELU_F16)[epi])
    finally:
        ops.gemm_set_tile(0)


def timeit(fn, n=6, w=2):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


cases = [(256, 256, 128), (256, 256, 192), (512, 512, 256), (300, 256, 640), (1297, 768, 768), (4096, 1280, 1280)]
if mode == "full":
    cases += [(20000, 1536, 128), (33000, 2304, 192), (4096, 5120, 1280), (4096, 1280, 5120), (70001, 768, 128), (65536, 1280, 1280)]
if TILE >= 16:
    cases = [(256, 128, 1280), (256, 256, 1280), (512, 384, 1280), (300, 256, 1344), (1297, 768, 768), (4096, 1280, 1280)]
    if mode == "full":
        cases += [(20000, 1536, 1152), (4096, 5120, 1280), (4096, 1280, 5120), (70001, 640, 1088), (65536, 1280, 1280)]
bad = 0
for (M, N, K) in cases:
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(dev).half()
    w = (torch.randn(N, K, generator=g) * 0.05).to(dev).half()
    bias = (torch.randn(N, generator=g) * 0.5).to(dev)
    resid = torch.randn(M, N, generator=g).to(dev)
    gamma = torch.randn(N, generator=g).to(dev)
    for epi in (0, 1, 2):
        variants = [(bias, resid, gamma)] if epi < 2 else [(bias, resid, gamma), (None, resid, None), (bias, None, None)]
        for (b_, r_, g_) in variants:
            ref_tile = 11 if (K >= 768 and N % 256 == 0) else 1
            o_ref = run(ref_tile, a, w, b_, epi, r_, g_)
            o_asm = run(TILE, a, w, b_, epi, r_, g_)
            torch.cuda.synchronize()
            same = torch.equal(o_ref, o_asm)
            d = (o_ref.float() - o_asm.float()).abs().max().item()
            nan = int(torch.isnan(o_asm.float()).sum().item())
            print(f"{M}x{N}x{K} epi{epi} bias={b_ is not None} resid={r_ is not None} gamma={g_ is not None}: "
                  f"bit-identical={same} maxdiff={d:.3e} nan={nan}", flush=True)
            tol = (2e-3 if epi < 2 else 2e-4) * (4 if TILE == 16 else 1) * max(1.0, o_ref.float().abs().max().item() / 8)
            if not same and d > tol:
                bad += 1
print("FAILED cases:", bad, flush=True)

if bench:
    shapes = [(65536, 3840, 1280, 0), (65536, 1280, 1280, 2), (65536, 5120, 1280, 1), (65536, 1280, 5120, 2), (8192, 8192, 8192, 0),
              (4096, 3840, 1280, 0), (4096, 5120, 1280, 1), (4096, 1280, 5120, 2), (20752, 2304, 768, 0), (20752, 3072, 768, 1), (32768, 5120, 1280, 1)]
    for (M, N, K, epi) in shapes:
        a = torch.randn(M, K, device=dev).half(); w = (torch.randn(N, K, device=dev) * 0.05).half()
        bias = torch.randn(N, device=dev); gamma = torch.randn(N, device=dev)
        out = torch.empty(M, N, device=dev, dtype=torch.float32 if epi == 2 else torch.float16)
        if epi == 2:
            out.normal_()
        res = []
        for rep in range(2):
            for tl in (11, 15, 16):
                ops.gemm_set_tile(tl)
                if epi == 2:
                    t = timeit(lambda: ops.gemm(a, w, bias, out=out, epilogue=ops.EPI_F32, resid=out))
                else:
                    t = timeit(lambda: ops.gemm(a, w, bias, out=out, epilogue=(ops.EPI_F16, ops.EPI_GELU_F16)[epi]))
                res.append(f"t{tl}={2*M*N*K/t/1e12:6.0f}")
            ops.gemm_set_tile(0)
            o16 = out if epi != 2 else torch.empty(M, N, device=dev, dtype=torch.float16)
            t = timeit(lambda: torch.matmul(a, w.t(), out=o16))
            res.append(f"blaslt={2*M*N*K/t/1e12:6.0f}")
        print(f"{M}x{N}x{K} epi{epi}: " + " ".join(res), flush=True)
