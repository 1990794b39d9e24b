#!/usr/bin/env python3
"""Shader clock and socket power (amdgpu hwmon, 20 ms period) while ONE kernel type runs back to back for ~1.5 s: which kernels of
the pipeline are power-capped, and at which clock they run. Shapes: the 16-slice SAM ViT-H launches.
  python tools/kernel_power.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib.util
import torch
from protosam_amd import ops
spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
dev = torch.device("cuda:0")
B = 16


def gemm_case(M, N, K, epi):
    a = torch.randn(M, K, device=dev).half(); w = (torch.randn(N, K, device=dev) * 0.05).half(); bias = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if epi == 2 else torch.float16)
    if epi == 2:
        out.normal_()
        return (lambda: ops.gemm(a, w, bias, out=out, epilogue=ops.EPI_F32, resid=out)), 2.0 * M * N * K
    return (lambda: ops.gemm(a, w, bias, out=out, epilogue=(ops.EPI_F16, ops.EPI_GELU_F16)[epi])), 2.0 * M * N * K


def attn_global():
    H, hd, N = 16, 80, 4096
    qkv = torch.randn(B, N, 3, H, hd, device=dev).half()
    rh = torch.randn(B, H, N, 64, device=dev) * 0.5; rw = torch.randn(B, H, N, 64, device=dev) * 0.5
    out = torch.empty(B, N, H * hd, device=dev, dtype=torch.float16)
    return (lambda: ops.attention(qkv, B, N, H, hd, hd ** -0.5, out=out, mode=1, rel_h=rh, rel_w=rw, gh=64, gw=64)), 4.0 * B * H * N * N * hd


def attn_window():
    H, hd, N, ws = 16, 80, 4096, 14
    qkv = torch.randn(B, N, 3, H, hd, device=dev).half(); pad = torch.randn(3, H, hd, device=dev).half()
    rp = ops.pack_rel_tables(torch.randn(2 * ws - 1, hd, device=dev) * 0.3, torch.randn(2 * ws - 1, hd, device=dev) * 0.3, True, hd)
    out = torch.empty(B, N, H * hd, device=dev, dtype=torch.float16)
    return (lambda: ops.attention(qkv, B, N, H, hd, hd ** -0.5, out=out, mode=2, rpack=rp, pad_row=pad, gh=64, gw=64, ws=ws)), 4.0 * B * H * 25 * 196 * 196 * hd


def attn_dino():
    H, hd, N = 12, 64, 1297
    qkv = torch.randn(B, N, 3, H, hd, device=dev).half()
    out = torch.empty(B, N, H * hd, device=dev, dtype=torch.float16)
    return (lambda: ops.attention(qkv, B, N, H, hd, hd ** -0.5, out=out)), 4.0 * B * H * N * N * hd


cases = [("qkv 65536x3840x1280 fp16", lambda: gemm_case(65536, 3840, 1280, 0)), ("fc1 65536x5120x1280 GELU", lambda: gemm_case(65536, 5120, 1280, 1)),
         ("fc2 65536x1280x5120 fp32+res", lambda: gemm_case(65536, 1280, 5120, 2)), ("proj 65536x1280x1280 fp32+res", lambda: gemm_case(65536, 1280, 1280, 2)),
         ("global attention hd80 N4096", attn_global), ("window attention hd80", attn_window), ("DINOv2 attention hd64 N1297", attn_dino)]
for name, mk in cases:
    fn, flops = mk()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    n = max(int(1.5e3 / max(e0.elapsed_time(e1), 1e-3)), 5)
    ps = bench.PowerSampler(0)
    time.sleep(0.3)
    ps.start()
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    pc = ps.stop()
    us = e0.elapsed_time(e1) / n * 1e3
    print(f"{name}: {us:8.1f} us per call, {flops / us / 1e6:6.0f} TFLOP/s | {pc and (pc['avg_w'], pc['max_w'])} W, sclk avg {pc and pc['sclk_mhz_avg']} min {pc and pc['sclk_mhz_min']} max {pc and pc['sclk_mhz_max']} MHz ({pc and pc['samples']} samples)", flush=True)
    del fn
    torch.cuda.empty_cache()
