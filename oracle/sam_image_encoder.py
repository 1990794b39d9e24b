"""Oracle: SAM ViTDet image encoder (fp32, CPU). Test infrastructure only (see oracle/__init__.py).

Follows models/segment_anything/modeling/image_encoder.py (ImageEncoderViT.forward :108-122,
Block.forward :174-193, Attention.forward :235-251, window_partition :254-277, window_unpartition
:280-300, get_rel_pos :303-334, add_decomposed_rel_pos :337-372, PatchEmbed :375-406) and
modeling/common.py (MLPBlock :13-26, LayerNorm2d :31-43). Parameters are read from a state dict with
the reference's key names under a prefix (e.g. ``image_encoder.``).
"""
import math

import torch
import torch.nn.functional as F

VIT_CFGS = {  # build_sam.py:14-44
    "vit_b": dict(embed_dim=768, depth=12, num_heads=12, global_attn_indexes=(2, 5, 8, 11)),
    "vit_l": dict(embed_dim=1024, depth=24, num_heads=16, global_attn_indexes=(5, 11, 17, 23)),
    "vit_h": dict(embed_dim=1280, depth=32, num_heads=16, global_attn_indexes=(7, 15, 23, 31)),
}
WINDOW = 14      # build_sam.py:78
PATCH = 16       # build_sam.py:63
LN_EPS = 1e-6    # build_sam.py:72 (partial(LayerNorm, eps=1e-6)); LayerNorm2d default eps, common.py:32


def rel_pos_table(q_size, k_size, rel_pos):
    """image_encoder.py:303-334. Rows of the (possibly linearly resized) table picked per (q,k) offset."""
    max_rel = int(2 * max(q_size, k_size) - 1)
    if rel_pos.shape[0] != max_rel:
        rp = F.interpolate(rel_pos.reshape(1, rel_pos.shape[0], -1).permute(0, 2, 1), size=max_rel, mode="linear")
        rp = rp.reshape(-1, max_rel).permute(1, 0)
    else:
        rp = rel_pos
    qc = torch.arange(q_size)[:, None] * max(k_size / q_size, 1.0)
    kc = torch.arange(k_size)[None, :] * max(q_size / k_size, 1.0)
    idx = (qc - kc) + (k_size - 1) * max(q_size / k_size, 1.0)
    return rp[idx.long()]  # [q, k, C]


def decomposed_rel_pos_terms(q, rel_pos_h, rel_pos_w, hw):
    """image_encoder.py:358-366: rel_h[b,h,w,kh], rel_w[b,h,w,kw] from the UNSCALED q [Bh, h*w, C]."""
    gh, gw = hw
    Rh = rel_pos_table(gh, gh, rel_pos_h)
    Rw = rel_pos_table(gw, gw, rel_pos_w)
    rq = q.reshape(q.shape[0], gh, gw, q.shape[-1])
    rel_h = torch.einsum("bhwc,hkc->bhwk", rq, Rh)
    rel_w = torch.einsum("bhwc,wkc->bhwk", rq, Rw)
    return rel_h, rel_w


def attention(x, sd, pre, num_heads, use_rel_pos=True):
    """image_encoder.py:235-251 on x [B', h, w, C] (a whole map or a batch of windows)."""
    Bp, gh, gw, C = x.shape
    hd = C // num_heads
    qkv = F.linear(x, sd[pre + "qkv.weight"], sd[pre + "qkv.bias"])
    qkv = qkv.reshape(Bp, gh * gw, 3, num_heads, hd).permute(2, 0, 3, 1, 4).reshape(3, Bp * num_heads, gh * gw, hd)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = (q * hd ** -0.5) @ k.transpose(-2, -1)
    if use_rel_pos:
        rel_h, rel_w = decomposed_rel_pos_terms(q, sd[pre + "rel_pos_h"], sd[pre + "rel_pos_w"], (gh, gw))
        attn = (attn.view(-1, gh, gw, gh, gw) + rel_h[..., :, None] + rel_w[..., None, :]).view(-1, gh * gw, gh * gw)
    attn = attn.softmax(dim=-1)
    o = (attn @ v).view(Bp, num_heads, gh, gw, hd).permute(0, 2, 3, 1, 4).reshape(Bp, gh, gw, C)
    return F.linear(o, sd[pre + "proj.weight"], sd[pre + "proj.bias"])


def window_partition(x, ws):
    """image_encoder.py:254-277: zero-pad to a multiple of ws, split into [B*nW, ws, ws, C]."""
    B, H, W, C = x.shape
    ph, pw = (ws - H % ws) % ws, (ws - W % ws) % ws
    if ph or pw:
        x = F.pad(x, (0, 0, 0, pw, 0, ph))
    Hp, Wp = H + ph, W + pw
    x = x.view(B, Hp // ws, ws, Wp // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws, ws, C)
    return x, (Hp, Wp)


def window_unpartition(w, ws, pad_hw, hw):
    """image_encoder.py:280-300."""
    Hp, Wp = pad_hw
    H, W = hw
    B = w.shape[0] // (Hp * Wp // ws // ws)
    x = w.view(B, Hp // ws, Wp // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, -1)
    return x[:, :H, :W, :]


def block(x, sd, pre, num_heads, window_size):
    """image_encoder.py:174-193."""
    C = x.shape[-1]
    y = F.layer_norm(x, (C,), sd[pre + "norm1.weight"], sd[pre + "norm1.bias"], LN_EPS)
    if window_size > 0:
        H, W = y.shape[1], y.shape[2]
        y, pad_hw = window_partition(y, window_size)
    y = attention(y, sd, pre + "attn.", num_heads)
    if window_size > 0:
        y = window_unpartition(y, window_size, pad_hw, (H, W))
    x = x + y
    z = F.layer_norm(x, (C,), sd[pre + "norm2.weight"], sd[pre + "norm2.bias"], LN_EPS)
    z = F.linear(z, sd[pre + "mlp.lin1.weight"], sd[pre + "mlp.lin1.bias"])
    z = F.gelu(z)  # nn.GELU() default = erf form (common.py:22)
    z = F.linear(z, sd[pre + "mlp.lin2.weight"], sd[pre + "mlp.lin2.bias"])
    return x + z


def layer_norm_2d(x, w, b, eps=LN_EPS):
    """common.py:38-43 (channel LayerNorm on NCHW, biased variance)."""
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return w[:, None, None] * x + b[:, None, None]


def image_encoder(x, sd, pre="image_encoder.", model_type="vit_b", depth=None, taps=None):
    """image_encoder.py:108-122. x [B,3,1024,1024] already normalised/padded -> [B,256,64,64].
    `depth` may truncate the block stack (used by small golden fixtures); `taps` collects intermediates."""
    cfg = VIT_CFGS[model_type]
    nh = cfg["num_heads"]
    nblk = cfg["depth"] if depth is None else depth
    x = F.conv2d(x, sd[pre + "patch_embed.proj.weight"], sd[pre + "patch_embed.proj.bias"], stride=PATCH)
    x = x.permute(0, 2, 3, 1)
    if pre + "pos_embed" in sd:
        x = x + sd[pre + "pos_embed"]
    if taps is not None:
        taps["tokens0"] = x.clone()
    for i in range(nblk):
        ws = 0 if i in cfg["global_attn_indexes"] else WINDOW
        x = block(x, sd, f"{pre}blocks.{i}.", nh, ws)
        if taps is not None:
            taps[f"block{i}"] = x.clone()
    x = x.permute(0, 3, 1, 2)
    x = F.conv2d(x, sd[pre + "neck.0.weight"])
    x = layer_norm_2d(x, sd[pre + "neck.1.weight"], sd[pre + "neck.1.bias"])
    x = F.conv2d(x, sd[pre + "neck.2.weight"], padding=1)
    x = layer_norm_2d(x, sd[pre + "neck.3.weight"], sd[pre + "neck.3.bias"])
    return x
