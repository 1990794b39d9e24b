#!/usr/bin/env python3
"""Window attention (SAM ViT-H shape, 16 slices), default dispatch: microseconds per call over 600 back-to-back calls + sampled clock."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib.util
import torch
from protosam_amd import ops
spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
dev = torch.device("cuda:0")
B, H, hd, N, ws = 16, 16, 80, 4096, 14
qkv = torch.randn(B, N, 3, H, hd, device=dev).half(); pad = torch.randn(3, H, hd, device=dev).half()
rp = ops.pack_rel_tables(torch.randn(2 * ws - 1, hd, device=dev) * 0.3, torch.randn(2 * ws - 1, hd, device=dev) * 0.3, True, hd)
out = torch.empty(B, N, H * hd, device=dev, dtype=torch.float16)
fn = lambda: ops.attention(qkv, B, N, H, hd, hd ** -0.5, out=out, mode=2, rpack=rp, pad_row=pad, gh=64, gw=64, ws=ws)
for _ in range(5):
    fn()
torch.cuda.synchronize()
ps = bench.PowerSampler(0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = int(os.environ.get("NCALLS", "2000"))
time.sleep(0.2); ps.start(); e0.record()
for _ in range(n):
    fn()
e1.record(); torch.cuda.synchronize()
pc = ps.stop()
us = e0.elapsed_time(e1) / n * 1e3
print(f"{os.environ.get('ABL', '?'):12s} window attention {us:7.1f} us per call, sclk {pc and pc['sclk_mhz_avg']} MHz, {pc and pc['avg_w']} W")
