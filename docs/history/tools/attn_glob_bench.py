import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protosam_amd import ops
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
H, hd, N = 16, 80, 4096
qkv = torch.randn(B, N, 3, H, hd, device=dev).half()
rh = torch.randn(B, H, N, 64, device=dev) * 0.5
rw = torch.randn(B, H, N, 64, device=dev) * 0.5
out = torch.empty(B, N, H * hd, device=dev, dtype=torch.float16)
def run(): ops.attention(qkv, B, N, H, hd, hd ** -0.5, out=out, mode=1, rel_h=rh, rel_w=rw, gh=64, gw=64)
for _ in range(2): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): run()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 5 * 1e-3
print(f"dbg={os.environ.get('PSAM_ATTN_DBG','0')} B={B}: {t*1e6:.0f} us  {4*B*H*N*N*hd/t/1e12:.0f} TF/s")
