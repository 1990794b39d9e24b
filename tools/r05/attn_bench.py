"""Round 5: the assembly global-attention kernels against what they replace, one process, interleaved.
  norel: psam_gattn_asm_64_norel vs gattn_kernel<64,0,false> (DINOv2-B/14: 16 x 1297 tokens, 1 x 1297, 1 x 5330; 12 heads)
  fused: psam_gattn_asm_80_fused vs psam_relpos + psam_gattn_asm_80_rel (SAM ViT-H global block: 16 slices and one)
python tools/r05/attn_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from protosam_amd import ops
dev = torch.device("cuda:0")


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (B, N, H) in ((16, 1297, 12), (1, 1297, 12), (2, 1297, 12), (1, 5330, 12), (4, 5330, 12)):
    hd = 64
    qkv = torch.randn(B, N, 3, H, hd, device=dev).half()
    out = torch.empty(B, N, H * hd, device=dev, dtype=torch.float16)
    fl = 4.0 * B * H * N * N * hd
    res = []
    for rep in range(2):
        for name, var in (("asm", 5), ("hip", 5 | 16)):
            ops.attention_set_variant(var)
            t = timeit(lambda: ops.attention(qkv, B, N, H, hd, hd ** -0.5, out=out))
            res.append((name, t))
    ops.attention_set_variant(5)
    print(f"norel B={B} N={N} H={H}: " + "  ".join(f"{n} {t:.1f} us ({fl / t / 1e6:.0f} TF/s)" for n, t in res), flush=True)

for (B, H, hd) in ((16, 16, 80), (1, 16, 80), (16, 12, 64)):
    N, g = 4096, 64
    qkv = torch.randn(B, N, 3, H, hd, device=dev).half()
    out = torch.empty(B, N, H * hd, device=dev, dtype=torch.float16)
    Rh, Rw = torch.randn(127, hd, device=dev) * 0.3, torch.randn(127, hd, device=dev) * 0.3
    rpack = ops.pack_rel_tables(Rh, Rw, False, hd)
    rh = torch.empty(B, H, N, 64, device=dev)
    rw = torch.empty(B, H, N, 64, device=dev)
    fl = 4.0 * B * H * N * N * hd

    def two():
        ops.relpos(qkv, rpack, B, N, H, hd, g, g, False, hd ** -0.5, rel_h=rh, rel_w=rw)
        ops.attention(qkv, B, N, H, hd, hd ** -0.5, out=out, mode=1, rel_h=rh, rel_w=rw, gh=g, gw=g)

    def rel_only():
        ops.attention(qkv, B, N, H, hd, hd ** -0.5, out=out, mode=1, rel_h=rh, rel_w=rw, gh=g, gw=g)

    def fused():
        ops.attention(qkv, B, N, H, hd, hd ** -0.5, out=out, mode=1, rpack=rpack, gh=g, gw=g)
    res = []
    for rep in range(2):
        for name, fn in (("fused", fused), ("relpos+rel", two), ("rel alone", rel_only)):
            res.append((name, timeit(fn, 5)))
    print(f"global rel-pos B={B} H={H} hd={hd}: " + "  ".join(f"{n} {t:.0f} us ({fl / t / 1e6:.0f} TF/s)" for n, t in res), flush=True)
