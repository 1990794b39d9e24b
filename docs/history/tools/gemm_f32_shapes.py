"""psam_gemm_f32 (the decoder's exact-fp32 GEMM) at decoder shapes: time per call; PSAM_GEMM_F32_SHAPE=0/1/2 forces 128x128 / 64x128 / 64x64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protosam_amd import ops
dev = torch.device("cuda:0")
for P in (16, 26, 40, 1, 2):
    for (N, K) in ((128, 256), (256, 128), (256, 256)):
        M = P * 4096
        a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
        out = torch.empty(M, N, device=dev)
        fn = lambda: ops.gemm_f32(a, w, b, out=out)
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"shape {os.environ.get('PSAM_GEMM_F32_SHAPE', 'auto'):4s} P={P:3d} M={M:7d} N={N} K={K}: {us:7.1f} us  {2.0 * M * N * K / us / 1e6:6.1f} TFLOP/s", flush=True)
