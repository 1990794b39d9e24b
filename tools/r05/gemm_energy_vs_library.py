#!/usr/bin/env python3
"""Round 5: the same instruments on our GEMM and on the vendor library, same tensors, one process - wall time, shader
clock and socket power (amdgpu hwmon, 20 ms period) while ONE launch type runs back to back for ~1.5 s, hence joules per TFLOP.
Rows per SAM ViT-H shape at 16 slices: ours as the pipeline launches it (bias / GELU / fp32 residual epilogue), ours with the plain fp16
epilogue and no bias (the closest thing to "no epilogue" the entry point offers), torch.matmul (hipBLASLt, no epilogue).
  python tools/r05/gemm_energy_vs_library.py > profiles/r05_gemm_energy_vs_library.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import importlib.util
import torch
from protosam_amd import ops
spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
dev = torch.device("cuda:0")


def run(name, fn, flops):
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
    tf = flops / us / 1e6
    if pc:
        jt = pc["avg_w"] * us * 1e-6 / (flops * 1e-12)
        print(f"  {name:58s} {us:7.1f} us {tf:6.0f} TFLOP/s | {pc['avg_w']:6.0f} W avg, sclk {pc['sclk_mhz_avg']} MHz | {jt:5.2f} J per TFLOP", flush=True)
    else:
        print(f"  {name:58s} {us:7.1f} us {tf:6.0f} TFLOP/s | (no hwmon files on this box)", flush=True)


print("# python tools/r05/gemm_energy_vs_library.py on 1x MI355X: Gaussian activations, 0.05-scaled Gaussian weights, one launch type back to back for ~1.5 s")
for (nm, M, N, K, epi) in (("qkv", 65536, 3840, 1280, 0), ("proj", 65536, 1280, 1280, 2), ("fc1", 65536, 5120, 1280, 1), ("fc2", 65536, 1280, 5120, 2)):
    a = torch.randn(M, K, device=dev).half()
    w = (torch.randn(N, K, device=dev) * 0.05).half()
    bias = torch.randn(N, device=dev)
    fl = 2.0 * M * N * K
    print(f"{nm}: {M} x {N} x {K}")
    if epi == 2:
        x = torch.randn(M, N, device=dev)
        x16 = torch.empty(M, N, device=dev, dtype=torch.float16)
        stats = torch.empty(M, N // 64, 2, device=dev)
        run("ours, fp32 residual epilogue (x += a w^T + b)", lambda: ops.gemm(a, w, bias, out=x, epilogue=ops.EPI_F32, resid=x), fl)
        run("ours, + folded-LayerNorm producer (fp16 copy, row sums)", lambda: ops.gemm(a, w, bias, out=x, epilogue=ops.EPI_F32, resid=x, out16=x16, stats=stats), fl)
    else:
        o = torch.empty(M, N, device=dev, dtype=torch.float16)
        run("ours, %s epilogue" % ("bias + fp16" if epi == 0 else "bias + GELU + fp16"), lambda: ops.gemm(a, w, bias, out=o, epilogue=epi), fl)
    o16 = torch.empty(M, N, device=dev, dtype=torch.float16)
    run("ours, plain fp16 store, no bias", lambda: ops.gemm(a, w, None, out=o16, epilogue=ops.EPI_F16), fl)
    run("torch.matmul (hipBLASLt), fp16 store, no epilogue", lambda: torch.matmul(a, w.t(), out=o16), fl)
    del a, w
    torch.cuda.empty_cache()
