"""Micro-timings of the core kernels on the GPU box (development aid, not the judged bench)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protosam_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, n=20, w=3):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


def main():
    for (M, N, K) in [(4096, 3840, 1280), (4096, 1280, 1280), (4096, 5120, 1280), (4096, 1280, 5120),
                      (4096, 2304, 768), (4096, 3072, 768), (4096, 768, 3072), (1297, 2304, 768),
                      (8 * 1297, 2304, 768), (8 * 1297, 3072, 768), (8192, 8192, 8192)]:
        a = torch.randn(M, K, device=dev).half()
        w = (torch.randn(N, K, device=dev) * 0.05).half()
        b = torch.randn(N, device=dev)
        out = torch.empty(M, N, device=dev, dtype=torch.float16)
        for tile in (1, 2, 3, 0):
            if tile == 3 and N % 256:
                continue
            ops.gemm_set_tile(tile)
            t = timeit(lambda: ops.gemm(a, w, b, out=out, epilogue=ops.EPI_F16))
            print(f"gemm {M}x{N}x{K} tile{tile}: {t*1e6:8.1f} us  {2*M*N*K/t/1e12:7.1f} TF/s", flush=True)
        ops.gemm_set_tile(0)
        tt = timeit(lambda: torch.matmul(a, w.t()))
        print(f"   torch(hipblaslt) same shape: {tt*1e6:8.1f} us  {2*M*N*K/tt/1e12:7.1f} TF/s", flush=True)
    for (M, D) in [(4096, 1280), (8 * 1297, 768)]:
        x = torch.randn(M, D, device=dev)
        w = torch.randn(D, device=dev); b = torch.randn(D, device=dev)
        y = torch.empty(M, D, device=dev, dtype=torch.float16)
        t = timeit(lambda: ops.layernorm(x, w, b, 1e-6, out=y))
        print(f"layernorm {M}x{D}: {t*1e6:8.1f} us  {(M*D*6)/t/1e9:7.1f} GB/s", flush=True)
    for (B, N, H, hd, mode) in [(1, 4096, 16, 80, 0), (1, 4096, 12, 64, 0), (8, 1297, 12, 64, 0), (1, 4096, 16, 80, 2),
                                (1, 4096, 16, 80, 1)]:
        qkv = torch.randn(B, N, 3, H, hd, device=dev).half()
        kw = {}
        if mode:
            K = 14 if mode == 2 else 64
            Rh = torch.randn(2 * K - 1, hd, device=dev) * 0.1
            Rw = torch.randn(2 * K - 1, hd, device=dev) * 0.1
            rp = ops.pack_rel_tables(Rh, Rw, mode == 2, hd)
            tr = timeit(lambda: ops.relpos(qkv, rp, B, N, H, hd, 64, K, mode == 2, hd ** -0.5))
            print(f"relpos mode{mode}: {tr*1e6:8.1f} us", flush=True)
            r = ops.relpos(qkv, rp, B, N, H, hd, 64, K, mode == 2, hd ** -0.5)
            if mode == 2:
                kw = dict(mode=2, relq=r, gh=64, gw=64, ws=14, pad_row=torch.randn(3, H, hd, device=dev).half())
            else:
                kw = dict(mode=1, rel_h=r[0], rel_w=r[1], gh=64, gw=64)
        out = torch.empty(B, N, H * hd, device=dev, dtype=torch.float16)
        t = timeit(lambda: ops.attention(qkv, B, N, H, hd, hd ** -0.5, out=out, **kw))
        nk = 196 if mode == 2 else N
        fl = 4.0 * B * H * N * nk * hd
        print(f"attention B{B} N{N} H{H} hd{hd} mode{mode}: {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF/s", flush=True)


if __name__ == "__main__":
    main()
