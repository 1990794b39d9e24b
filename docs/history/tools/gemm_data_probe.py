"""Data dependence of the GEMM rate (the chip is power-managed: operand bit toggling moves the clock): tiles 11 / 14 / 13 on
zero, uniform [-1,1) and Gaussian operands at 4096^3 and 8192^3 (the guide's 8-phase template quotes uniform-random numbers)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protosam_amd import ops
dev = torch.device("cuda:0")


def timed(fn, n=10):
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


for S in (4096, 8192):
    for name, mk in (("zeros", lambda s: torch.zeros(s, device=dev)), ("uniform[-1,1)", lambda s: torch.rand(s, device=dev) * 2 - 1),
                     ("randn", lambda s: torch.randn(s, device=dev)), ("randn x 0.05 weights", None)):
        if mk is None:
            a, w = torch.randn((S, S), device=dev).half(), (torch.randn((S, S), device=dev) * 0.05).half()
        else:
            a, w = mk((S, S)).half(), mk((S, S)).half()
        o = torch.empty((S, S), device=dev, dtype=torch.float16)
        res = []
        for tile in (11, 14, 13, 11):
            ops.gemm_set_tile(tile)
            us = timed(lambda: ops.gemm(a, w, None, out=o, epilogue=ops.EPI_F16))
            res.append(f"tile {tile}: {2.0 * S ** 3 / us / 1e6:.0f}")
        ops.gemm_set_tile(0)
        # the library GEMM torch dispatches to (hipBLASLt / rocBLAS) on the SAME tensors, before and after
        wt = w.t()
        us = timed(lambda: torch.matmul(a, wt, out=o))
        res.append(f"torch.matmul: {2.0 * S ** 3 / us / 1e6:.0f}")
        ops.gemm_set_tile(11)
        us = timed(lambda: ops.gemm(a, w, None, out=o, epilogue=ops.EPI_F16))
        ops.gemm_set_tile(0)
        res.append(f"tile 11 again: {2.0 * S ** 3 / us / 1e6:.0f}")
        print(f"{S}^3 {name}: " + "  ".join(res) + " TFLOP/s")


# the benchmark's own shapes (16 slices), Gaussian operands: ours (automatic tile) against the library
for M, N, K in ((65536, 3840, 1280), (65536, 5120, 1280), (65536, 1280, 5120), (65536, 1280, 1280)):
    a, w = torch.randn((M, K), device=dev).half(), (torch.randn((N, K), device=dev) * 0.05).half()
    o = torch.empty((M, N), device=dev, dtype=torch.float16)
    wt = w.t()
    r = []
    for rep in range(2):
        us = timed(lambda: ops.gemm(a, w, None, out=o, epilogue=ops.EPI_F16))
        r.append(f"ours {2.0 * M * N * K / us / 1e6:.0f}")
        us = timed(lambda: torch.matmul(a, wt, out=o))
        r.append(f"torch.matmul {2.0 * M * N * K / us / 1e6:.0f}")
    print(f"{M}x{N}x{K} fp16 out, no bias: " + "  ".join(r) + " TFLOP/s")
