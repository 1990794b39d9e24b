"""Round 5: error of the global-attention kernels against an fp64 evaluation of the same fp16 operands (one launch each): the fused
rel-pos kernel, the two-kernel path it replaces (psam_relpos -> _rel kernel), the HIP kernel; DINOv2's no-bias kernels likewise.
A shift of the error LEVEL would show here; a reshuffle of which output rounds which way does not.
  python tools/r05/attn_accuracy.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from protosam_amd import ops
dev = torch.device("cuda:0")


def ref64(qkv, B, N, H, hd, scale, bias=None):
    q, k, v = qkv.double().view(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
    att = (q * scale) @ k.transpose(-2, -1)
    if bias is not None:
        att = att + bias
    return (att.softmax(-1) @ v).permute(0, 2, 1, 3).reshape(B, N, H * hd)


def stats(name, out, ref):
    e = (out.double() - ref).abs()
    print(f"  {name:28s} max {e.max().item():.3e}  rms {e.pow(2).mean().sqrt().item():.3e}  mean {e.mean().item():.3e}", flush=True)


for hd in (80, 64):
    B, H, g = 1, 2, 64
    N = g * g
    gen = torch.Generator().manual_seed(7)
    qkv = (torch.randn((B, N, 3, H, hd), generator=gen)).to(dev).half()
    Rh = (torch.randn((127, hd), generator=gen) * 0.3).to(dev)
    Rw = (torch.randn((127, hd), generator=gen) * 0.3).to(dev)
    scale = hd ** -0.5
    q = qkv.double().view(B, N, 3, H, hd)[:, :, 0].permute(0, 2, 1, 3)          # [B,H,N,hd]
    idx = torch.arange(g, device=dev)[:, None] - torch.arange(g, device=dev)[None, :] + (g - 1)
    rh = torch.einsum("bhyxc,ykc->bhyxk", q.view(B, H, g, g, hd), Rh.double()[idx])      # [B,H,qy,qx,ky]
    rw = torch.einsum("bhyxc,xkc->bhyxk", q.view(B, H, g, g, hd), Rw.double()[idx])      # [B,H,qy,qx,kx]
    bias = (rh[..., :, None] + rw[..., None, :]).reshape(B, H, N, N)
    ref = ref64(qkv, B, N, H, hd, scale, bias)
    rpack = ops.pack_rel_tables(Rh, Rw, False, hd)
    print(f"global + rel-pos, hd = {hd}: |out| rms {ref.pow(2).mean().sqrt().item():.3f}")
    fused = ops.attention(qkv, B, N, H, hd, scale, mode=1, rpack=rpack, gh=g, gw=g)
    stats("fused (asm)", fused, ref)
    rel_h, rel_w = ops.relpos(qkv, rpack, B, N, H, hd, g, g, False, scale)
    print(f"  psam_relpos vs fp64: rel_h max {(rel_h.double().view(B, H, g, g, g) - rh).abs().max().item():.2e}  rel_w max {(rel_w.double().view(B, H, g, g, g) - rw).abs().max().item():.2e}")
    two = ops.attention(qkv, B, N, H, hd, scale, mode=1, rel_h=rel_h, rel_w=rel_w, gh=g, gw=g)
    stats("psam_relpos + _rel (asm)", two, ref)
    ops.attention_set_variant(5 | 16)
    hip = ops.attention(qkv, B, N, H, hd, scale, mode=1, rel_h=rel_h, rel_w=rel_w, gh=g, gw=g)
    ops.attention_set_variant(5)
    stats("psam_relpos + gattn (HIP)", hip, ref)
    stats("fp16 rounding of the fp64 result", ref.half(), ref)
    d = (fused.double() - two.double()).abs()
    print(f"  fused vs two-kernel: max {d.max().item():.3e} rms {d.pow(2).mean().sqrt().item():.3e}; per query-block-of-256 max: "
          + " ".join(f"{d.view(B, 16, 256, -1)[0, i].max().item():.1e}" for i in range(16)))
for N in (1297, 5330):
    B, H, hd = 1, 12, 64
    qkv = torch.randn((B, N, 3, H, hd), generator=torch.Generator().manual_seed(9)).to(dev).half()
    ref = ref64(qkv, B, N, H, hd, hd ** -0.5)
    print(f"no bias, N = {N}")
    stats("norel (asm)", ops.attention(qkv, B, N, H, hd, hd ** -0.5), ref)
    ops.attention_set_variant(5 | 16)
    stats("gattn (HIP)", ops.attention(qkv, B, N, H, hd, hd ** -0.5), ref)
    ops.attention_set_variant(5)
    stats("fp16 rounding of the fp64 result", ref.half(), ref)
