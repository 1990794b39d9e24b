#!/usr/bin/env python3
"""Within-process interleaved A/B of the attention kernel's softmax variants (guide rule 24): global + rel-pos (SAM ViT-H
shape), global without bias (DINOv2 shape), window + rel-pos. Prints median / min microseconds and TFLOP/s per variant."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from protosam_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 7
variants = [int(v) for v in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["0", "1"])]


def timed(fn, n=3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def case_global_rel():
    H, hd, N = 16, 80, 4096
    qkv = torch.randn(B, N, 3, H, hd, device=dev).half()
    rh = torch.randn(B, H, N, 64, device=dev) * 0.5
    rw = torch.randn(B, H, N, 64, device=dev) * 0.5
    out = torch.empty(B, N, H * hd, device=dev, dtype=torch.float16)
    return (lambda: ops.attention(qkv, B, N, H, hd, hd ** -0.5, out=out, mode=1, rel_h=rh, rel_w=rw, gh=64, gw=64)), \
        4.0 * B * H * N * N * hd


def case_dino():
    H, hd, N = 12, 64, 1297
    qkv = torch.randn(B, N, 3, H, hd, device=dev).half()
    out = torch.empty(B, N, H * hd, device=dev, dtype=torch.float16)
    return (lambda: ops.attention(qkv, B, N, H, hd, hd ** -0.5, out=out)), 4.0 * B * H * N * N * hd


def case_window():
    H, hd, N, ws = 16, 80, 4096, 14
    qkv = torch.randn(B, N, 3, H, hd, device=dev).half()
    pad = torch.randn(3, H, hd, device=dev).half()
    rp = ops.pack_rel_tables(torch.randn(2 * ws - 1, hd, device=dev) * 0.3, torch.randn(2 * ws - 1, hd, device=dev) * 0.3,
                             True, hd)
    out = torch.empty(B, N, H * hd, device=dev, dtype=torch.float16)
    return (lambda: ops.attention(qkv, B, N, H, hd, hd ** -0.5, out=out, mode=2, rpack=rp, pad_row=pad, gh=64, gw=64,
                                  ws=ws)), 4.0 * B * H * 25 * 196 * 196 * hd


for name, mk in (("global+relpos hd80 N4096", case_global_rel), ("global hd64 N1297", case_dino),
                 ("window+relpos hd80", case_window)):
    fn, flops = mk()
    res = {v: [] for v in variants}
    for v in variants:
        ops.attention_set_variant(v)
        fn()
    torch.cuda.synchronize()
    for _ in range(ROUNDS):
        for v in variants:
            ops.attention_set_variant(v)
            res[v].append(timed(fn))
    for v in variants:
        med, mn = statistics.median(res[v]), min(res[v])
        print(f"{name} B={B} variant {v}: median {med:.0f} us  min {mn:.0f} us  {flops / med / 1e6:.0f} TFLOP/s")
ops.attention_set_variant(5)
