"""A/B of the assembly GEMM's experiment variants (library built with `make -C protosam_amd/csrc GENFLAGS=--experiments`):
  PSAM_GEMM_ASM_TRACE=1 python tools/gemm_asm_ab.py 0,1,2,... [MxNxKxEPI;...]
variant 0 = shipped schedule, t11 = the HIP persistent kernel; the trace lines (stderr) give cycles per K-tile / per epilogue."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protosam_amd import ops
dev = torch.device("cuda:0")
variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,1").split(",")]
shapes = sys.argv[2] if len(sys.argv) > 2 else "65536x3840x1280x0;65536x5120x1280x1;65536x1280x5120x2;65536x1280x1280x2;8192x8192x8192x0"


def timeit(fn, n=5, w=2):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


for sh in shapes.split(";"):
    M, N, K, epi = (int(v) for v in sh.split("x"))
    a = torch.randn(M, K, device=dev).half(); w = (torch.randn(N, K, device=dev) * 0.05).half()
    bias = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if epi == 2 else torch.float16)
    if epi == 2:
        out.normal_()
    fn = (lambda: ops.gemm(a, w, bias, out=out, epilogue=ops.EPI_F32, resid=out)) if epi == 2 else \
         (lambda: ops.gemm(a, w, bias, out=out, epilogue=(ops.EPI_F16, ops.EPI_GELU_F16)[epi]))
    res = []
    for rep in range(2):
        ops.gemm_set_tile(11)
        res.append(f"t11={2*M*N*K/timeit(fn)/1e12:5.0f}")
        ops.gemm_set_tile(int(os.environ.get('TILE', '15')))
        for v in variants:
            ops.gemm_asm_variant(v)
            os.environ["PSAM_TRACE_QUIET"] = "1"
            t = timeit(fn, n=3 if v > 0 else 5, w=1)
            res.append(f"v{v}={2*M*N*K/t/1e12:5.0f}")
        ops.gemm_asm_variant(0)
        ops.gemm_set_tile(0)
    print(f"{sh}: " + " ".join(res), flush=True)
