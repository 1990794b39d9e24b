"""LayerNorm pass at the SAM ViT-H shapes: microseconds and GB/s (fp32 in, fp16 out)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protosam_amd import ops
dev = torch.device("cuda:0")
for M, D in ((65536, 1280), (4096, 1280), (20752, 768)):
    x = torch.randn(M, D, device=dev)
    w, b = torch.randn(D, device=dev), torch.randn(D, device=dev)
    y = torch.empty(M, D, device=dev, dtype=torch.float16)
    f = lambda: ops.layernorm(x, w, b, 1e-6, out=y)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"layernorm {M}x{D}: {us:.1f} us  {M * D * 6 / us / 1e3:.0f} GB/s")
