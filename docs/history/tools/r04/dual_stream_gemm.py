"""Do two half-batch streams of GEMMs on half the CUs each beat one full-batch stream? (epilogue bursts of one beside the k-loops of
the other).  python tools/r04/dual_stream_gemm.py [layers]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from protosam_amd import ops
dev = torch.device("cuda:0")
L = int(sys.argv[1]) if len(sys.argv) > 1 else 8
D, H4 = 1280, 5120


class Chain:
    def __init__(self, M):
        g = lambda *s: torch.randn(*s, device=dev)
        self.M = M
        self.x = g(M, D); self.x16 = self.x.half(); self.stats = torch.empty(M, D // 64, 2, device=dev)
        self.qkv = torch.empty(M, 3 * D, device=dev, dtype=torch.float16)
        self.att = g(M, D).half()
        self.h = torch.empty(M, H4, device=dev, dtype=torch.float16)
        self.wqkv = (g(3 * D, D) * 0.03).half(); self.bqkv = g(3 * D) * 0.1
        self.wproj = (g(D, D) * 0.03).half(); self.bproj = g(D) * 0.1
        self.w1 = (g(H4, D) * 0.03).half(); self.b1 = g(H4) * 0.1
        self.w2 = (g(D, H4) * 0.01).half(); self.b2 = g(D) * 0.1
        self.ops = [self.f_qkv, self.f_proj, self.f_fc1, self.f_fc2]

    def f_qkv(self): ops.gemm(self.x16, self.wqkv, self.bqkv, out=self.qkv, epilogue=ops.EPI_F16)
    def f_proj(self): ops.gemm(self.att, self.wproj, self.bproj, out=self.x, epilogue=ops.EPI_F32, resid=self.x, out16=self.x16, stats=self.stats)
    def f_fc1(self): ops.gemm(self.x16, self.w1, self.b1, out=self.h, epilogue=ops.EPI_GELU_F16)
    def f_fc2(self): ops.gemm(self.h, self.w2, self.b2, out=self.x, epilogue=ops.EPI_F32, resid=self.x, out16=self.x16, stats=self.stats)


def run_single(c, n):
    for _ in range(n):
        for f in c.ops: f()


def run_dual(c1, c2, s1, s2, n, shift):
    for i in range(n * 4):
        with torch.cuda.stream(s1):
            c1.ops[i % 4]()
        with torch.cuda.stream(s2):
            c2.ops[(i + shift) % 4]()


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


flops = lambda M: L * 2.0 * M * (3 * D * D + D * D + 2 * D * H4)
full = Chain(65536)
t = timed(lambda: run_single(full, L))
print(f"single stream, M=65536: {t / L * 1e3:8.1f} us per layer of GEMMs, {flops(65536) / t / 1e9:7.0f} TFLOP/s", flush=True)
h1, h2 = Chain(32768), Chain(32768)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for cap in (0, 128):
    ops.gemm_set_option("max_wgs", cap)
    for shift in (0, 1, 2):
        def f():
            s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
            run_dual(h1, h2, s1, s2, L, shift)
            torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
        t = timed(f)
        print(f"two streams, M=32768 each, cap {cap:3d}, phase shift {shift}: {t / L * 1e3:8.1f} us per layer pair, {2 * flops(32768) / t / 1e9:7.0f} TFLOP/s", flush=True)
    t = timed(lambda: (run_single(h1, L), run_single(h2, L)))
    print(f"one stream, two M=32768 chains back to back, cap {cap:3d}: {t / L * 1e3:8.1f} us, {2 * flops(32768) / t / 1e9:7.0f} TFLOP/s", flush=True)
ops.gemm_set_option("max_wgs", 0)
