#!/usr/bin/env python3
"""CPU emulation (build container): what would folding the block LayerNorms into the consuming GEMMs cost in accuracy?

Scheme A (current HIP path): A operand = fp16(LayerNorm(x) * gamma + beta), W operand = fp16(W).
Scheme B (LN folded):        A operand = fp16(x) (the raw fp32 residual stream, rounded), W operand = fp16(W * gamma);
                             y = rstd * (acc - mean * s) + t  with s[n] = sum_k W'[n,k], t[n] = bias[n] + sum_k beta[k] W[n,k],
                             mean / rstd from fp32 sum(x), sum(x^2).
Both with fp32 accumulation; proj / lin2 and the neck use fp16-rounded operands in both schemes; attention in fp32.
Prints the image-embedding error of each scheme against the all-fp32 oracle on a seeded ViT-B (or --type vit_h --depth N).
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import sam_image_encoder as oenc  # noqa: E402

r16 = lambda t: t.half().float()  # noqa: E731


def lin16(x, w, b):
    return F.linear(r16(x), r16(w), b)


def ln_lin(x, sd, npre, lpre, scheme):
    C = x.shape[-1]
    g, beta = sd[npre + "weight"], sd[npre + "bias"]
    W, b = sd[lpre + "weight"], sd[lpre + "bias"]
    if scheme == "A":
        return lin16(F.layer_norm(x, (C,), g, beta, oenc.LN_EPS), W, b)
    Wp = r16(W * g[None, :])
    s = Wp.sum(1)
    t = b + W @ beta
    mean = x.sum(-1, keepdim=True) / C
    var = (x * x).sum(-1, keepdim=True) / C - mean * mean
    rstd = torch.rsqrt(var + oenc.LN_EPS)
    acc = F.linear(r16(x), Wp)
    return rstd * (acc - mean * s) + t


def attention_from_qkv(qkv, sd, pre, num_heads, gh, gw):
    Bp = qkv.shape[0]
    C = qkv.shape[-1] // 3
    hd = C // num_heads
    qkv = qkv.reshape(Bp, gh * gw, 3, num_heads, hd).permute(2, 0, 3, 1, 4).reshape(3, Bp * num_heads, gh * gw, hd)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = (q * hd ** -0.5) @ k.transpose(-2, -1)
    rel_h, rel_w = oenc.decomposed_rel_pos_terms(q, sd[pre + "rel_pos_h"], sd[pre + "rel_pos_w"], (gh, gw))
    attn = (attn.view(-1, gh, gw, gh, gw) + rel_h[..., :, None] + rel_w[..., None, :]).view(-1, gh * gw, gh * gw)
    o = (attn.softmax(-1) @ v).view(Bp, num_heads, gh, gw, hd).permute(0, 2, 3, 1, 4).reshape(Bp, gh, gw, C)
    return lin16(o, sd[pre + "proj.weight"], sd[pre + "proj.bias"])


def block(x, sd, pre, nh, ws, scheme):
    qkv = ln_lin(x, sd, pre + "norm1.", pre + "attn.qkv.", scheme)          # [B,64,64,3C]
    if ws > 0:
        H, W = qkv.shape[1], qkv.shape[2]
        bias = sd[pre + "attn.qkv.bias"]
        w, pad_hw = oenc.window_partition(qkv - bias, ws)                   # zero-padded LN output -> qkv = bias there
        y = attention_from_qkv(w + bias, sd, pre + "attn.", nh, ws, ws)
        y = oenc.window_unpartition(y, ws, pad_hw, (H, W))
    else:
        y = attention_from_qkv(qkv, sd, pre + "attn.", nh, qkv.shape[1], qkv.shape[2])
    x = x + y
    z = F.gelu(ln_lin(x, sd, pre + "norm2.", pre + "mlp.lin1.", scheme))
    return x + lin16(z, sd[pre + "mlp.lin2.weight"], sd[pre + "mlp.lin2.bias"])


def encoder(x, sd, model_type, depth, scheme):
    cfg = oenc.VIT_CFGS[model_type]
    pre = "image_encoder."
    x = F.conv2d(r16(x), r16(sd[pre + "patch_embed.proj.weight"]), sd[pre + "patch_embed.proj.bias"], stride=16)
    x = x.permute(0, 2, 3, 1) + sd[pre + "pos_embed"]
    for i in range(depth):
        ws = 0 if i in cfg["global_attn_indexes"] else 14
        x = block(x, sd, f"{pre}blocks.{i}.", cfg["num_heads"], ws, scheme)
    x = x.permute(0, 3, 1, 2)
    x = F.conv2d(r16(x), r16(sd[pre + "neck.0.weight"]))
    x = oenc.layer_norm_2d(x, sd[pre + "neck.1.weight"], sd[pre + "neck.1.bias"])
    x = F.conv2d(r16(x), r16(sd[pre + "neck.2.weight"]), padding=1)
    return oenc.layer_norm_2d(x, sd[pre + "neck.3.weight"], sd[pre + "neck.3.bias"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--type", default="vit_b")
    ap.add_argument("--depth", type=int, default=12)
    ap.add_argument("--mean-shift", type=float, default=0.0, help="adds a per-token common-mode offset to the residual stream "
                    "(stress: fp16(x) loses what LayerNorm would have removed)")
    args = ap.parse_args()
    from protosam_amd.segment_anything import sam_model_registry
    from protosam_amd.synth import synth_pair, synth_state_dict
    torch.set_num_threads(os.cpu_count())
    sd = synth_state_dict(sam_model_registry[args.type](encoder_depth=args.depth), 1234)
    if args.mean_shift:
        sd["image_encoder.pos_embed"] = sd["image_encoder.pos_embed"] + args.mean_shift
    _, _, q, _ = synth_pair(1024, seed=1)
    q = (q - q.min()) / (q.max() - q.min()) * 255
    x = (q.to(torch.uint8).float() - torch.tensor([123.675, 116.28, 103.53]).view(1, 3, 1, 1)) / torch.tensor(
        [58.395, 57.12, 57.375]).view(1, 3, 1, 1)
    with torch.no_grad():
        ref = oenc.image_encoder(x, sd, model_type=args.type, depth=args.depth)
        for scheme in ("A", "B"):
            out = encoder(x, sd, args.type, args.depth, scheme)
            e = (out - ref).abs()
            print(f"scheme {scheme}: embedding max abs err {e.max():.3e}, mean {e.mean():.3e} (ref rms {ref.pow(2).mean().sqrt():.3f})")


if __name__ == "__main__":
    main()
