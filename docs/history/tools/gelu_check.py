"""GELU epilogue of every GEMM tile against torch's erf GELU (fp32) and against tile 1 bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protosam_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(5)
for (M, N, K) in ((256, 256, 64), (1297, 768, 768), (4096, 1280, 1280)):
    a = (torch.randn((M, K), generator=g)).half().to(dev); w = (torch.randn((N, K), generator=g) * 0.05).half().to(dev)
    bias = (torch.randn((N,), generator=g) * 0.5).to(dev)
    ref = torch.nn.functional.gelu(a.float() @ w.float().t() + bias)
    outs = {}
    for tile in (1, 11, 15, 16):
        ops.gemm_set_tile(tile)
        outs[tile] = ops.gemm(a, w, bias, epilogue=ops.EPI_GELU_F16).float()
        ops.gemm_set_tile(0)
    for tile, o in outs.items():
        err = (o - ref).abs()
        print(f"{M}x{N}x{K} tile {tile}: max abs err vs torch gelu {err.max().item():.3e} (mean {err.mean().item():.2e}), differs from tile 1 in {(o != outs[1]).sum().item()} of {o.numel()}")
