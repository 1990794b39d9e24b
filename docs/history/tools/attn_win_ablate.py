"""In-process ablation sweep of the window-attention kernel (PSAM_ATTN_DBG bits, csrc/attention.hip wattn_kernel: 1 no K/V DMA,
2 no rel-pos prologue, 4 no key chunks, 8 no softmax, 16 no PV, 32 no one-hot loads, 64 no query loads). Times only; results of
ablated runs are meaningless."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protosam_amd import ops
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
masks = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1, 2, 4, 8, 16, 32, 64, 24, 28, 7, 127, 0]
H, hd, N = 16, 80, 4096
qkv = torch.randn(B, N, 3, H, hd, device=dev).half()
pad = torch.randn(3, H, hd, device=dev).half()
out = torch.empty(B, N, H * hd, device=dev, dtype=torch.float16)
rp = ops.pack_rel_tables(torch.randn(27, hd, device=dev) * 0.3, torch.randn(27, hd, device=dev) * 0.3, True, hd)


def run():
    ops.attention(qkv, B, N, H, hd, hd ** -0.5, out=out, mode=2, rpack=rp, gh=64, gw=64, ws=14, pad_row=pad)


for variant in ([int(v) for v in os.environ.get("VARIANTS", "3").split(",")]):
    ops.attention_set_variant(variant)
    for m in masks:
        os.environ["PSAM_ATTN_DBG"] = str(m)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        print(f"variant {variant} dbg {m:3d} B={B}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us")
os.environ["PSAM_ATTN_DBG"] = "0"
