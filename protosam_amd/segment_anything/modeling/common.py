"""Modules shared by the SAM models (names as models/segment_anything/modeling/common.py). On the hot path their arithmetic runs
inside other kernels' epilogues (ImageEncoderViT._encode_patches, TwoWayAttentionBlock._run); the `forward`s here are thin
drivers of the same kernels for callers that use the modules directly."""
import torch
import torch.nn as nn

from ... import ops


class MLPBlock(nn.Module):
    """common.py:13-26: lin2(act(lin1(x)))."""

    def __init__(self, embedding_dim, mlp_dim, act=nn.GELU):
        super().__init__()
        self.lin1 = nn.Linear(embedding_dim, mlp_dim)
        self.lin2 = nn.Linear(mlp_dim, embedding_dim)
        self.act = act()

    @torch.no_grad()
    def forward(self, x):
        """x [..., embedding_dim] on the device -> fp32 [..., embedding_dim]. GELU with GEMM-sized widths (the image encoder's blocks):
        fp16-operand MFMA GEMMs, the activation in the first one's epilogue (`psam_gemm_f16`, epilogues 1 and 2); ReLU (the two-way
        transformer's block, 256 -> 2048 -> 256 on a few token rows): fp32 `psam_small_linear`."""
        D = x.shape[-1]
        x2 = x.reshape(-1, D)
        l1, l2 = self.lin1, self.lin2
        if isinstance(self.act, nn.GELU) and getattr(self.act, "approximate", "none") == "none" and D % 64 == 0 and l1.out_features % 128 == 0 \
                and D % 128 == 0:
            hid = ops.gemm(x2.half().contiguous(), f16(l1.weight), f32(l1.bias), epilogue=ops.EPI_GELU_F16)
            out = ops.gemm(hid, f16(l2.weight), f32(l2.bias), epilogue=ops.EPI_F32)
        elif isinstance(self.act, nn.ReLU):
            hid = ops.small_linear(x2.float().contiguous(), f32(l1.weight), f32(l1.bias), act=1)
            out = ops.small_linear(hid, f32(l2.weight), f32(l2.bias))
        else:
            raise NotImplementedError(f"MLPBlock.forward: activation {type(self.act).__name__} / widths {D} -> {l1.out_features} have "
                                      "no HIP kernel (GELU with widths that are multiples of 128, or ReLU)")
        return out.view(tuple(x.shape[:-1]) + (l2.out_features,))


class LayerNorm2d(nn.Module):
    """common.py:31-43: LayerNorm over the channels of an NCHW map (eps 1e-6)."""

    def __init__(self, num_channels, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(num_channels))
        self.bias = nn.Parameter(torch.zeros(num_channels))
        self.eps = eps

    @torch.no_grad()
    def forward(self, x):
        """x [B,C,H,W] on the device -> fp32 [B,C,H,W] (a permuted view of the token-major rows `psam_layernorm` works on: the
        biased variance over C and `(x - u) / sqrt(s + eps)` of :39-42 are the row LayerNorm of the [B*H*W, C] matrix)."""
        B, C, H, W = x.shape
        rows = x.permute(0, 2, 3, 1).reshape(B * H * W, C).float().contiguous()
        y = ops.layernorm(rows, f32(self.weight), f32(self.bias), self.eps, out_dtype=torch.float32)
        return y.view(B, H, W, C).permute(0, 3, 1, 2)


def f32(p):
    return p.detach().float().contiguous()


def f16(p):
    return p.detach().half().contiguous()
