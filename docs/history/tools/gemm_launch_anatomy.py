"""Wall time per launch (HIP events over n back-to-back launches) beside the traced cycles of the same launch (experiment variant 1 of
the assembly kernels: k-loop, epilogue, kernel entry -> exit), for one-slice shapes.
  PSAM_GEMM_ASM_CO=build/gemm_asm_exp.co PSAM_GEMM_ASM_TRACE=1 python tools/gemm_launch_anatomy.py [tile=15] [MxNxKxEPI;...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protosam_amd import ops
dev = torch.device("cuda:0")
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 15
shapes = sys.argv[2] if len(sys.argv) > 2 else "4096x3840x1280x0;4096x1280x1280x2;4096x5120x1280x1;4096x1280x5120x2"
for sh in shapes.split(";"):
    M, N, K, epi = (int(v) for v in sh.split("x"))
    a = torch.randn(M, K, device=dev).half(); w = (torch.randn(N, K, device=dev) * 0.05).half()
    bias = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if epi == 2 else torch.float16)
    if epi == 2:
        out.normal_()
    fn = (lambda: ops.gemm(a, w, bias, out=out, epilogue=ops.EPI_F32, resid=out)) if epi == 2 else \
         (lambda: ops.gemm(a, w, bias, out=out, epilogue=(ops.EPI_F16, ops.EPI_GELU_F16)[epi]))
    ops.gemm_set_tile(tile)
    ops.gemm_asm_variant(0)
    for _ in range(5): fn()
    torch.cuda.synchronize()
    n = 50
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    print(f"tile {tile} {sh}: {us:.1f} us per launch back to back ({2 * M * N * K / us / 1e6:.0f} TFLOP/s)", flush=True)
    sys.stdout.flush()
    ops.gemm_asm_variant(1)
    fn(); fn()
    torch.cuda.synchronize()
    ops.gemm_asm_variant(0)
    ops.gemm_set_tile(0)
