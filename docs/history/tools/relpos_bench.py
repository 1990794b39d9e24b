"""psam_relpos (global form) at the SAM ViT-H shape: microseconds per call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protosam_amd import ops
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
H, hd, N, g = 16, 80, 4096, 64
qkv = torch.randn(B, N, 3, H, hd, device=dev).half()
rp = ops.pack_rel_tables(torch.randn(2 * g - 1, hd, device=dev) * 0.3, torch.randn(2 * g - 1, hd, device=dev) * 0.3, False, hd)
rh = torch.empty((B, H, N, 64), dtype=torch.float32, device=dev)
rw = torch.empty_like(rh)
f = lambda: ops.relpos(qkv, rp, B, N, H, hd, g, g, False, hd ** -0.5, rel_h=rh, rel_w=rw)
for _ in range(3):
    f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    f()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 10 * 1e3
print(f"relpos global B={B}: {us:.0f} us ({2 * rh.numel() * 4 / us / 1e3:.0f} GB/s of output)")
