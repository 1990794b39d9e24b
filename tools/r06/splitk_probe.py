"""Does a K-split of the one-slice fc2 GEMM (4096 x 1280 x 5120, 80 tiles of 256 x 256 for 256 CUs) as THREE concurrent launches of the
256-tile assembly kernel on K ranges (three streams, 240 workgroups side by side) beat the half-tile kernel? Times the k-loops only
(plain fp32 stores to three planes); the reduce + LayerNorm pass is priced separately by its bytes."""
import sys
import torch
sys.path.insert(0, ".")
from protosam_amd import ops

dev = torch.device("cuda:0")
M, N, K = 4096, 1280, 5120
g = torch.Generator().manual_seed(0)
a = (torch.randn((M, K), generator=g)).half().to(dev)
w = (torch.randn((N, K), generator=g) * 0.02).half().to(dev)
b = torch.randn(N, generator=g).to(dev)
x = torch.randn((M, N), generator=g).to(dev)
lnw, lnb = torch.ones(N, device=dev), torch.zeros(N, device=dev)


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


out = x.clone()
ops.gemm(a, w, b, out=out, epilogue=ops.EPI_F32, resid=out)
torch.cuda.synchronize()
g0 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g0):
    for _ in range(10):
        ops.gemm(a, w, b, out=out, epilogue=ops.EPI_F32, resid=out)
base = timeit(g0.replay, 20) / 10
ln16 = torch.empty((M, N), dtype=torch.float16, device=dev)
ln = timeit(lambda: ops.layernorm(out, lnw, lnb, 1e-6, out=ln16))
print(f"current: fc2 one slice {base:.1f} us + LayerNorm pass {ln:.1f} us")
for ks in (2, 3, 4):
    kt = K // 64
    cuts = [round(i * kt / ks) * 64 for i in range(ks + 1)]
    planes = torch.empty((ks, M, N), dtype=torch.float32, device=dev)
    streams = [torch.cuda.Stream() for _ in range(ks)]
    ops.gemm_set_tile(15)

    def run():
        cur = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(cur)
        for r in range(ks):
            streams[r].wait_event(ev)
            with torch.cuda.stream(streams[r]):
                ops.gemm(a[:, cuts[r]:cuts[r + 1]], w[:, cuts[r]:cuts[r + 1]], None, out=planes[r], epilogue=ops.EPI_F32)
        for r in range(ks):
            cur.wait_stream(streams[r])
    run()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(10):
            run()
    t = timeit(gr.replay, 20) / 10
    ops.gemm_set_tile(0)
    ref = a.float() @ w.float().t()
    err = (planes.sum(0) - ref).abs().max().item()
    red = timeit(lambda: torch.add(planes.sum(0), out))
    print(f"ks={ks}: {ks} concurrent tile-15 launches {t:.1f} us (sum check {err:.2e}); a torch sum + add of the planes {red:.1f} us "
          f"(bytes of a fused reduce + LayerNorm: {(ks + 1) * M * N * 4 + M * N * 6} B = {((ks + 1) * M * N * 4 + M * N * 6) / 5.0e6:.1f} us at 5 TB/s)")
