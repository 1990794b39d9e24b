"""Parameter containers shared by the SAM modules (names as models/segment_anything/modeling/common.py)."""
import torch
import torch.nn as nn


class MLPBlock(nn.Module):
    """common.py:13-26 (lin1 -> act -> lin2); arithmetic runs in the GEMM epilogues."""

    def __init__(self, embedding_dim, mlp_dim, act=nn.GELU):
        super().__init__()
        self.lin1 = nn.Linear(embedding_dim, mlp_dim)
        self.lin2 = nn.Linear(mlp_dim, embedding_dim)
        self.act = act()


class LayerNorm2d(nn.Module):
    """common.py:31-43: channel LayerNorm (eps 1e-6); applied on token-major rows by csrc/layernorm.hip."""

    def __init__(self, num_channels, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(num_channels))
        self.bias = nn.Parameter(torch.zeros(num_channels))
        self.eps = eps


def f32(p):
    return p.detach().float().contiguous()


def f16(p):
    return p.detach().half().contiguous()
