import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protosam_amd import ops
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
H, hd, N = 16, 80, 4096
torch.manual_seed(0)
qkv = torch.randn(B, N, 3, H, hd, device=dev).half()
r = torch.randn(B, H, N, 2, 32, device=dev).half() * 0.1
pad = torch.randn(3, H, hd, device=dev).half()
out = torch.empty(B, N, H * hd, device=dev, dtype=torch.float16)
Rh, Rw = torch.randn(27, hd, device=dev) * 0.3, torch.randn(27, hd, device=dev) * 0.3
rp = ops.pack_rel_tables(Rh, Rw, True, hd)
FUSED = os.environ.get("FUSED", "0") == "1"
def run():
    if FUSED:
        ops.attention(qkv, B, N, H, hd, hd ** -0.5, out=out, mode=2, rpack=rp, gh=64, gw=64, ws=14, pad_row=pad)
    else:
        ops.attention(qkv, B, N, H, hd, hd ** -0.5, out=out, mode=2, relq=r, gh=64, gw=64, ws=14, pad_row=pad)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
print(f"fused={int(FUSED)} dbg={os.environ.get('PSAM_ATTN_DBG','0')} B={B}: {e0.elapsed_time(e1)/20*1e3:.1f} us")
print("checksum", out.float().sum().item(), out.float().abs().sum().item(), out[B - 1, 1000, 77].item(), out[0, 4095, 1279].item(), int(torch.isnan(out.float()).sum()))
