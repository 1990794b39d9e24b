"""Global attention with rel-pos (SAM ViT-H shape) through the kernel PSAM_GATTN selects; saves / compares the output.
   PSAM_GATTN=1 python tools/attn_glob_check.py B save /tmp/ref.pt ;  PSAM_GATTN=3 python tools/attn_glob_check.py B cmp /tmp/ref.pt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protosam_amd import ops
dev = torch.device("cuda:0")
B = int(sys.argv[1]); mode = sys.argv[2]; path = sys.argv[3]
H, hd, N = 16, 80, 4096
torch.manual_seed(0)
qkv = torch.randn(B, N, 3, H, hd, device=dev).half()
rh = torch.randn(B, H, N, 64, device=dev) * 0.5
rw = torch.randn(B, H, N, 64, device=dev) * 0.5
out = torch.zeros(B, N, H * hd, device=dev, dtype=torch.float16)
def run(): ops.attention(qkv, B, N, H, hd, hd ** -0.5, out=out, mode=1, rel_h=rh, rel_w=rw, gh=64, gw=64)
run(); torch.cuda.synchronize()
if mode == "save":
    torch.save(out.cpu(), path)
else:
    ref = torch.load(path).float(); got = out.cpu().float()
    d = (ref - got).abs()
    print("nan", int(torch.isnan(got).sum()), "max abs diff", d.max().item(), "mean", d.mean().item(), "ref absmax", ref.abs().max().item())
    bad = (d > 5e-3).nonzero()
    print("bad elements", bad.shape[0], bad[:8].tolist())
for _ in range(2): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): run()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 5 * 1e-3
print(f"PSAM_GATTN={os.environ.get('PSAM_GATTN','1')} B={B}: {t*1e6:.0f} us  {4*B*H*N*N*hd/t/1e12:.0f} TF/s")
