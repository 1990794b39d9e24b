"""Oracle: DINOv2 ViT `forward_features` (fp32, CPU). Test infrastructure only (see oracle/__init__.py).

PARITY UNPINNED against the hub code: `facebookresearch/dinov2` is fetched by torch.hub in the reference
(models/grid_proto_fewshot.py:54-72) and is absent from /root/reference; there is no network. This is a
restatement of the published architecture (DinoVisionTransformer with NestedTensorBlock / Attention /
Mlp / LayerScale / PatchEmbed), cross-checked against the independent `transformers.Dinov2Model`
(`oracle/validate_against_reference.py`). The contract at the reference call site is
`encoder.forward_features(x)["x_norm_patchtokens"]` -> [B, (S//14)^2, D] (grid_proto_fewshot.py:90-91).
State-dict keys are the hub names: cls_token, pos_embed, (register_tokens), patch_embed.proj.*,
blocks.{i}.{norm1,norm2}.*, blocks.{i}.attn.{qkv,proj}.*, blocks.{i}.ls{1,2}.gamma,
blocks.{i}.mlp.fc{1,2}.*, norm.*.
"""
import math

import torch
import torch.nn.functional as F

DINO_CFGS = {
    "dinov2_b14": dict(embed_dim=768, depth=12, num_heads=12, num_register_tokens=0, interpolate_antialias=False,
                       interpolate_offset=0.1),
    "dinov2_l14": dict(embed_dim=1024, depth=24, num_heads=16, num_register_tokens=0, interpolate_antialias=False,
                       interpolate_offset=0.1),
    "dinov2_l14_reg": dict(embed_dim=1024, depth=24, num_heads=16, num_register_tokens=4,
                           interpolate_antialias=True, interpolate_offset=0.0),
}
PATCH = 14
LN_EPS = 1e-6


def _linear(sd, pre, x, lora_scale=1.0):
    """nn.Linear, or util/lora.py's LoraInjectedLinear (:34-59) when the state dict carries an injected layer
    (`<pre>linear.{weight,bias}`, `<pre>lora_down.weight` [r,in], `<pre>lora_up.weight` [out,r]; eval: dropout is the
    identity, selector is nn.Identity, scale = 1.0 as inject_trainable_lora's default, :265):
    linear(x) + lora_up(lora_down(x)) * scale."""
    if pre + "linear.weight" in sd:
        y = F.linear(x, sd[pre + "linear.weight"], sd.get(pre + "linear.bias"))
        return y + F.linear(F.linear(x, sd[pre + "lora_down.weight"]), sd[pre + "lora_up.weight"]) * lora_scale
    return F.linear(x, sd[pre + "weight"], sd[pre + "bias"])


def interpolate_pos_encoding(pos_embed, npatch_side_w, npatch_side_h, antialias=False, offset=0.1):
    """Bicubic resample of the (M x M) patch part of pos_embed to (w0 x h0); cls part untouched.
    With offset != 0 the hub passes scale_factor=((w0+offset)/M, (h0+offset)/M) (so the sampling
    scale is M/(w0+offset), not M/w0); with offset == 0 it passes size=(w0, h0)."""
    N = pos_embed.shape[1] - 1
    M = int(math.sqrt(N))
    w0, h0 = npatch_side_w, npatch_side_h
    if w0 * h0 == N and w0 == h0:
        return pos_embed
    dim = pos_embed.shape[-1]
    cls_pe = pos_embed[:, :1]
    grid = pos_embed[:, 1:].reshape(1, M, M, dim).permute(0, 3, 1, 2)
    if offset:
        kw = dict(scale_factor=(float(w0 + offset) / M, float(h0 + offset) / M))
    else:
        kw = dict(size=(w0, h0))
    grid = F.interpolate(grid, mode="bicubic", antialias=antialias, **kw)
    assert (w0, h0) == tuple(grid.shape[-2:])
    grid = grid.permute(0, 2, 3, 1).reshape(1, -1, dim)
    return torch.cat([cls_pe, grid], dim=1)


def attention(x, sd, pre, num_heads):
    B, N, C = x.shape
    hd = C // num_heads
    qkv = _linear(sd, pre + "qkv.", x).reshape(B, N, 3, num_heads, hd)
    qkv = qkv.permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * hd ** -0.5, qkv[1], qkv[2]
    a = (q @ k.transpose(-2, -1)).softmax(dim=-1)
    o = (a @ v).transpose(1, 2).reshape(B, N, C)
    return _linear(sd, pre + "proj.", o)


def block(x, sd, pre, num_heads):
    C = x.shape[-1]
    y = F.layer_norm(x, (C,), sd[pre + "norm1.weight"], sd[pre + "norm1.bias"], LN_EPS)
    x = x + sd[pre + "ls1.gamma"] * attention(y, sd, pre + "attn.", num_heads)
    y = F.layer_norm(x, (C,), sd[pre + "norm2.weight"], sd[pre + "norm2.bias"], LN_EPS)
    y = _linear(sd, pre + "mlp.fc1.", y)
    y = _linear(sd, pre + "mlp.fc2.", F.gelu(y))
    return x + sd[pre + "ls2.gamma"] * y


def forward_features(x, sd, which="dinov2_b14", pre="", depth=None, taps=None):
    """x [B,3,H,W] with H,W multiples of 14 -> dict(x_norm_clstoken, x_norm_patchtokens)."""
    cfg = DINO_CFGS[which]
    B, _, H, W = x.shape
    t = F.conv2d(x, sd[pre + "patch_embed.proj.weight"], sd[pre + "patch_embed.proj.bias"], stride=PATCH)
    t = t.flatten(2).transpose(1, 2)  # [B, n, D], row-major over (h, w)
    t = torch.cat([sd[pre + "cls_token"].expand(B, -1, -1), t], dim=1)
    t = t + interpolate_pos_encoding(sd[pre + "pos_embed"], W // PATCH, H // PATCH, cfg["interpolate_antialias"],
                                     cfg["interpolate_offset"])
    R = cfg["num_register_tokens"]
    if R:
        t = torch.cat([t[:, :1], sd[pre + "register_tokens"].expand(B, -1, -1), t[:, 1:]], dim=1)
    if taps is not None:
        taps["tokens0"] = t.clone()
    nblk = cfg["depth"] if depth is None else depth
    for i in range(nblk):
        t = block(t, sd, f"{pre}blocks.{i}.", cfg["num_heads"])
        if taps is not None:
            taps[f"block{i}"] = t.clone()
    C = t.shape[-1]
    t = F.layer_norm(t, (C,), sd[pre + "norm.weight"], sd[pre + "norm.bias"], LN_EPS)
    return {"x_norm_clstoken": t[:, 0], "x_norm_patchtokens": t[:, 1 + R:]}
