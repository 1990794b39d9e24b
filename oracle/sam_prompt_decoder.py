"""Oracle: SAM prompt encoder, two-way mask decoder and mask post-processing (fp32, CPU).
Test infrastructure only (see oracle/__init__.py).

Follows models/segment_anything/modeling/prompt_encoder.py (PromptEncoder.forward :128-168, _embed_points
:73-92, _embed_boxes :94-101, get_dense_pe :62-71, PositionEmbeddingRandom :171-214),
modeling/transformer.py (TwoWayTransformer.forward :62-106, TwoWayAttentionBlock.forward :151-182,
Attention.forward :218-240), modeling/mask_decoder.py (predict_masks :112-149, forward :71-110, MLP
:154-176), modeling/sam.py (postprocess_masks: `Sam` :133-161 nearest; `SamBatched` :292-321 bilinear
align_corners=True; upstream segment_anything 1.0 bilinear align_corners=False), predictor.py
(predict :92-167, predict_torch :169-241) and utils/transforms.py (apply_coords :40-52).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

EMBED = 256
IMG = 1024
GRID = 64


def pe_encoding(coords01, G):
    """prompt_encoder.py:186-193: coords in [0,1]^2 -> [sin, cos](2*pi*(2c-1)@G)."""
    c = (2 * coords01 - 1) @ G
    c = 2 * np.pi * c
    return torch.cat([torch.sin(c), torch.cos(c)], dim=-1)


def dense_pe(sd, pre="prompt_encoder."):
    """prompt_encoder.py:62-71,195-206 -> [1,256,64,64]."""
    G = sd[pre + "pe_layer.positional_encoding_gaussian_matrix"]
    ones = torch.ones((GRID, GRID), dtype=torch.float32)
    y = (ones.cumsum(dim=0) - 0.5) / GRID
    x = (ones.cumsum(dim=1) - 0.5) / GRID
    return pe_encoding(torch.stack([x, y], dim=-1), G).permute(2, 0, 1)[None]


def embed_with_coords(coords, G):
    """prompt_encoder.py:208-214 (coords in input-image pixels)."""
    c = coords.clone().to(torch.float)
    c[:, :, 0] = c[:, :, 0] / IMG
    c[:, :, 1] = c[:, :, 1] / IMG
    return pe_encoding(c, G)


def _ln2d(x, w, b, eps=1e-6):
    """common.py:31-43 (LayerNorm2d over the channel dim of NCHW)."""
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    return w[:, None, None] * ((x - u) / torch.sqrt(s + eps)) + b[:, None, None]


def embed_masks(sd, masks, pre="prompt_encoder."):
    """prompt_encoder.py:51-59,102-105: masks [B,1,256,256] -> [B,256,64,64]."""
    m = pre + "mask_downscaling."
    x = F.conv2d(masks.float(), sd[m + "0.weight"], sd[m + "0.bias"], stride=2)
    x = F.gelu(_ln2d(x, sd[m + "1.weight"], sd[m + "1.bias"]))
    x = F.conv2d(x, sd[m + "3.weight"], sd[m + "3.bias"], stride=2)
    x = F.gelu(_ln2d(x, sd[m + "4.weight"], sd[m + "4.bias"]))
    return F.conv2d(x, sd[m + "6.weight"], sd[m + "6.bias"])


def prompt_encoder(sd, points=None, boxes=None, pre="prompt_encoder.", masks=None):
    """points = (coords [B,N,2], labels [B,N]) or None; boxes [B,4] or None; masks [B,1,256,256] or None.
    Returns sparse [B,Ns,256], dense [B,256,64,64]."""
    G = sd[pre + "pe_layer.positional_encoding_gaussian_matrix"]
    bs = points[0].shape[0] if points is not None else (boxes.shape[0] if boxes is not None else
                                                        (masks.shape[0] if masks is not None else 1))
    sparse = torch.empty((bs, 0, EMBED))
    if points is not None:
        coords, labels = points
        coords = coords + 0.5
        if boxes is None:  # pad with a not-a-point (prompt_encoder.py:80-84,155)
            coords = torch.cat([coords, torch.zeros((bs, 1, 2))], dim=1)
            labels = torch.cat([labels, -torch.ones((bs, 1))], dim=1)
        pe = embed_with_coords(coords, G)
        pe[labels == -1] = 0.0
        pe[labels == -1] += sd[pre + "not_a_point_embed.weight"]
        pe[labels == 0] += sd[pre + "point_embeddings.0.weight"]
        pe[labels == 1] += sd[pre + "point_embeddings.1.weight"]
        sparse = torch.cat([sparse, pe], dim=1)
    if boxes is not None:
        b = (boxes + 0.5).reshape(-1, 2, 2)
        ce = embed_with_coords(b, G)
        ce[:, 0, :] += sd[pre + "point_embeddings.2.weight"]
        ce[:, 1, :] += sd[pre + "point_embeddings.3.weight"]
        sparse = torch.cat([sparse, ce], dim=1)
    if masks is not None:
        dense = embed_masks(sd, masks, pre)
    else:
        dense = sd[pre + "no_mask_embed.weight"].reshape(1, -1, 1, 1).expand(bs, -1, GRID, GRID)
    return sparse, dense


def _attn(sd, pre, q, k, v, num_heads=8):
    """transformer.py:218-240."""
    q = F.linear(q, sd[pre + "q_proj.weight"], sd[pre + "q_proj.bias"])
    k = F.linear(k, sd[pre + "k_proj.weight"], sd[pre + "k_proj.bias"])
    v = F.linear(v, sd[pre + "v_proj.weight"], sd[pre + "v_proj.bias"])

    def heads(x):
        b, n, c = x.shape
        return x.reshape(b, n, num_heads, c // num_heads).transpose(1, 2)

    q, k, v = heads(q), heads(k), heads(v)
    a = (q @ k.permute(0, 1, 3, 2)) / math.sqrt(q.shape[-1])
    o = torch.softmax(a, dim=-1) @ v
    b, h, n, c = o.shape
    o = o.transpose(1, 2).reshape(b, n, h * c)
    return F.linear(o, sd[pre + "out_proj.weight"], sd[pre + "out_proj.bias"])


def _ln(sd, pre, x):
    return F.layer_norm(x, (x.shape[-1],), sd[pre + "weight"], sd[pre + "bias"], 1e-5)  # nn.LayerNorm default eps


def two_way_transformer(sd, pre, src, pos_src, tokens):
    """transformer.py:62-106 with depth 2 blocks (:151-182)."""
    keys = src.flatten(2).permute(0, 2, 1)
    key_pe = pos_src.flatten(2).permute(0, 2, 1)
    queries, query_pe = tokens, tokens
    for i in range(2):
        lp = f"{pre}layers.{i}."
        if i == 0:  # skip_first_layer_pe
            queries = _attn(sd, lp + "self_attn.", queries, queries, queries)
        else:
            q = queries + query_pe
            queries = queries + _attn(sd, lp + "self_attn.", q, q, queries)
        queries = _ln(sd, lp + "norm1.", queries)
        q = queries + query_pe
        k = keys + key_pe
        queries = _ln(sd, lp + "norm2.", queries + _attn(sd, lp + "cross_attn_token_to_image.", q, k, keys))
        m = F.linear(F.relu(F.linear(queries, sd[lp + "mlp.lin1.weight"], sd[lp + "mlp.lin1.bias"])),
                     sd[lp + "mlp.lin2.weight"], sd[lp + "mlp.lin2.bias"])
        queries = _ln(sd, lp + "norm3.", queries + m)
        q = queries + query_pe
        k = keys + key_pe
        keys = _ln(sd, lp + "norm4.", keys + _attn(sd, lp + "cross_attn_image_to_token.", k, q, queries))
    q = queries + query_pe
    k = keys + key_pe
    queries = _ln(sd, pre + "norm_final_attn.", queries + _attn(sd, pre + "final_attn_token_to_image.", q, k, keys))
    return queries, keys


def _mlp3(sd, pre, x):
    for i in range(3):
        x = F.linear(x, sd[f"{pre}layers.{i}.weight"], sd[f"{pre}layers.{i}.bias"])
        if i < 2:
            x = F.relu(x)
    return x


def mask_decoder(sd, image_embeddings, image_pe, sparse, dense, multimask_output, pre="mask_decoder.", taps=None):
    """mask_decoder.py:71-149 -> (masks [B,3|1,256,256], iou [B,3|1])."""
    B = sparse.shape[0]
    out_tok = torch.cat([sd[pre + "iou_token.weight"], sd[pre + "mask_tokens.weight"]], dim=0)
    tokens = torch.cat([out_tok[None].expand(B, -1, -1), sparse], dim=1)
    src = torch.repeat_interleave(image_embeddings, B, dim=0) + dense
    pos_src = torch.repeat_interleave(image_pe, B, dim=0)
    b, c, h, w = src.shape
    hs, src2 = two_way_transformer(sd, pre + "transformer.", src, pos_src, tokens)
    iou_tok = hs[:, 0, :]
    mask_toks = hs[:, 1:5, :]
    src2 = src2.transpose(1, 2).view(b, c, h, w)
    up = F.conv_transpose2d(src2, sd[pre + "output_upscaling.0.weight"], sd[pre + "output_upscaling.0.bias"], stride=2)
    u = up.mean(1, keepdim=True)
    s = (up - u).pow(2).mean(1, keepdim=True)
    up = (up - u) / torch.sqrt(s + 1e-6)
    up = sd[pre + "output_upscaling.1.weight"][:, None, None] * up + sd[pre + "output_upscaling.1.bias"][:, None, None]
    up = F.gelu(up)
    up = F.gelu(F.conv_transpose2d(up, sd[pre + "output_upscaling.3.weight"], sd[pre + "output_upscaling.3.bias"],
                                   stride=2))
    hyper = torch.stack([_mlp3(sd, f"{pre}output_hypernetworks_mlps.{i}.", mask_toks[:, i, :]) for i in range(4)],
                        dim=1)
    b, c, h, w = up.shape
    masks = (hyper @ up.view(b, c, h * w)).view(b, -1, h, w)
    iou = _mlp3(sd, pre + "iou_prediction_head.", iou_tok)
    if taps is not None:
        taps.update(hs=hs, keys=src2, upscaled=up, hyper=hyper, masks_all=masks, iou_all=iou)
    sl = slice(1, None) if multimask_output else slice(0, 1)
    return masks[:, sl], iou[:, sl]


def postprocess_masks(masks, input_size, original_size, variant="upstream"):
    """low-res logits -> original-size logits. variant: 'upstream' (pip segment_anything 1.0: bilinear,
    align_corners=False), 'batched' (vendored SamBatched, sam.py:313-320: bilinear align_corners=True),
    'nearest' (vendored Sam, sam.py:154-160)."""
    if variant == "nearest":
        m = F.interpolate(masks, (IMG, IMG), mode="nearest")
        m = m[..., : input_size[0], : input_size[1]]
        return F.interpolate(m, original_size, mode="nearest")
    ac = variant == "batched"
    m = F.interpolate(masks, (IMG, IMG), mode="bilinear", align_corners=ac)
    m = m[..., : int(input_size[0]), : int(input_size[1])]
    return F.interpolate(m, original_size, mode="bilinear", align_corners=ac)


def apply_coords(coords, original_size, target=IMG):
    """utils/transforms.py:40-52,137-148 (float64 numpy, like the reference)."""
    oh, ow = original_size
    scale = target * 1.0 / max(oh, ow)
    nh, nw = int(oh * scale + 0.5), int(ow * scale + 0.5)
    c = np.array(coords, dtype=float, copy=True)
    c[..., 0] = c[..., 0] * (nw / ow)
    c[..., 1] = c[..., 1] * (nh / oh)
    return c


def predict(sd, features, point_coords, point_labels, box, multimask_output, original_size, variant="upstream",
            mask_input=None, input_size=(IMG, IMG)):
    """predictor.py:92-241 for one prompt set. Returns (masks bool [C,H,W], iou [C], low_res [C,256,256])."""
    pts = None
    if point_coords is not None:
        pc = torch.as_tensor(apply_coords(point_coords, original_size), dtype=torch.float)[None]
        pl = torch.as_tensor(point_labels, dtype=torch.int)[None]
        pts = (pc, pl)
    bx = None
    if box is not None:
        bx = torch.as_tensor(apply_coords(np.asarray(box).reshape(-1, 2, 2), original_size).reshape(-1, 4),
                             dtype=torch.float)
    mk = None
    if mask_input is not None:                                                   # predictor.py:158-160
        mk = torch.as_tensor(mask_input, dtype=torch.float)[None, :, :, :]
    sparse, dense = prompt_encoder(sd, pts, bx, masks=mk)
    low, iou = mask_decoder(sd, features, dense_pe(sd), sparse, dense, multimask_output)
    masks = postprocess_masks(low, input_size, original_size, variant)
    return (masks > 0.0)[0], iou[0], low[0]
