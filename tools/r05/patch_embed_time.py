"""SAM ViT-H patch embedding of 16 slices (65536 x 1280 x 768, residual = position table [4096, 1280], LayerNorm producer outputs):
the assembly 256-tile kernel (tile 15) against the HIP persistent kernel (tile 11) it replaces there.  python tools/r05/patch_embed_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from protosam_amd import ops
dev = torch.device("cuda:0")
M, N, K, mod = 65536, 1280, 768, 4096
a = torch.randn(M, K, device=dev).half(); w = (torch.randn(N, K, device=dev) * 0.05).half()
bias = torch.randn(N, device=dev); pos = torch.randn(mod, N, device=dev)
x = torch.empty(M, N, device=dev); x16 = torch.empty(M, N, device=dev, dtype=torch.float16); st = torch.empty(M, N // 64, 2, device=dev)
for rep in range(2):
    for tile in (15, 11):
        ops.gemm_set_tile(tile)
        f = lambda: ops.gemm(a, w, bias, out=x, epilogue=ops.EPI_F32, resid=pos, resid_mod=mod, out16=x16, stats=st)   # noqa: E731
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 20 * 1e3
        print(f"tile {tile}: {t:.1f} us ({2.0 * M * N * K / t / 1e6:.0f} TFLOP/s)")
ops.gemm_set_tile(0)
