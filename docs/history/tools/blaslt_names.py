import torch, sys
dev = torch.device("cuda:0")
for sh in (sys.argv[1] if len(sys.argv) > 1 else "8192x8192x8192;32768x3840x1280;32768x5120x1280;32768x1280x5120;32768x1280x1280;10376x2304x768").split(";"):
    M, N, K = (int(v) for v in sh.split("x"))
    a = torch.randn(M, K, device=dev).half(); w = torch.randn(N, K, device=dev).half()
    out = torch.empty(M, N, device=dev, dtype=torch.float16)
    for _ in range(3): torch.matmul(a, w.t(), out=out)
    torch.cuda.synchronize()
