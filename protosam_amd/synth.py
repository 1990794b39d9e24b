"""Seeded synthetic weights and inputs (no checkpoints or datasets exist offline).

Weights: every parameter/buffer is filled from a CPU generator seeded by (seed, crc32(name)) so the same
state dict is reproduced on any machine and in any construction order, then handed both to the product
modules (`load_state_dict`) and to the oracle. Scales are chosen so every term of the arithmetic is
exercised (non-zero rel-pos tables, pos-embeds, biases, LayerScale) with O(1) activations.
"""
import math
import zlib

import numpy as np
import torch


_EMBEDDING_TABLES = ("iou_token", "mask_tokens", "point_embeddings", "not_a_point_embed", "no_mask_embed")


def _gen(seed, name):
    g = torch.Generator()
    g.manual_seed((int(seed) * 1000003 + zlib.crc32(name.encode())) % (2 ** 63 - 1))
    return g


def synth_tensor(name, shape, seed):
    g = _gen(seed, name)
    shape = tuple(shape)
    last = name.rsplit(".", 1)[-1]
    n = lambda s: torch.randn(shape, generator=g) * s  # noqa: E731
    if "rel_pos" in last:
        return n(0.1)
    if last in ("pos_embed", "cls_token", "register_tokens"):
        return n(0.5)
    if last == "mask_token":
        return n(0.02)
    if last == "gamma":
        return torch.rand(shape, generator=g) * 0.5 + 0.5
    if last == "positional_encoding_gaussian_matrix":
        return n(1.0)
    if last == "running_var":
        return torch.rand(shape, generator=g) * 0.5 + 0.75   # BatchNorm statistics: strictly positive
    if last == "num_batches_tracked":
        return torch.zeros(shape)
    if last == "bias":
        return n(0.1)
    if last == "weight":
        if len(shape) == 1:
            return 1.0 + n(0.1)  # norm scales
        if any(k in name for k in _EMBEDDING_TABLES):
            return n(1.0)  # nn.Embedding default init
        fan_in = int(np.prod(shape[1:]))
        if "output_upscaling" in name:  # ConvTranspose2d weight is [in, out, kh, kw]
            fan_in = shape[0]
        return n(1.0 / math.sqrt(fan_in))
    return n(0.1)


def synth_state_dict(module, seed=1234):
    """A full state dict for `module` (parameters and persistent buffers), fp32 on CPU."""
    sd = {}
    for k, v in module.state_dict().items():
        sd[k] = synth_tensor(k, v.shape, seed).to(torch.float32)
    return sd


def heavy_tail_sam_(sd, seed=1234, parts=("scale", "massive", "student")):
    """In place: give a synthetic SAM state dict the residual-stream statistics of trained ViT checkpoints (stress fixture for
    the fp16 operand path; tests/test_fullsize_gpu.py): per-channel scales spread over ~1.5 decades on what enters the stream
    (patch embedding, position embedding), a few 'massive activation' channels (tens everywhere, hundreds at a handful of token
    positions), and heavy-tailed (Student-t, 3 degrees of freedom) weights in every block's output projections."""
    g = _gen(seed, "heavy_tail")
    D = sd["image_encoder.pos_embed"].shape[-1]
    scale = torch.exp(torch.randn(D, generator=g)).clamp(0.25, 8.0)
    if "scale" in parts:
        sd["image_encoder.patch_embed.proj.weight"] *= scale[:, None, None, None]
        sd["image_encoder.patch_embed.proj.bias"] *= scale
        sd["image_encoder.pos_embed"] *= scale
    chans = torch.randperm(D, generator=g)[:3]
    gh, gw = sd["image_encoder.pos_embed"].shape[1:3]
    for c in (chans.tolist() if "massive" in parts else []):
        sign = 1.0 if c % 2 == 0 else -1.0
        sd["image_encoder.pos_embed"][..., c] += sign * 30.0
        for _ in range(6):
            y, x = int(torch.randint(0, gh, (1,), generator=g)), int(torch.randint(0, gw, (1,), generator=g))
            sd["image_encoder.pos_embed"][0, y, x, c] += sign * 250.0
    for k in (sd if "student" in parts else []):
        if k.startswith("image_encoder.blocks.") and (k.endswith("attn.proj.weight") or k.endswith("mlp.lin2.weight")):
            w = sd[k]
            chi2 = (torch.randn((3, w.shape[0]), generator=g) ** 2).sum(0)      # (seeded: 3 degrees of freedom per output channel)
            t = torch.randn(w.shape, generator=g) / torch.sqrt(chi2[:, None] / 3.0)
            sd[k] = (t / math.sqrt(w.shape[1]) * 0.7).to(torch.float32)
    return sd


# ---- synthetic slices / volumes ---------------------------------------------------------------
def _smooth_field(size, seed, n_blobs=12):
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32) / size
    f = np.zeros((size, size), np.float32)
    for _ in range(n_blobs):
        cy, cx = rng.uniform(0.1, 0.9, 2)
        s = rng.uniform(0.05, 0.25)
        a = rng.uniform(-1.0, 1.0)
        f += a * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * s * s))
    return f


def ellipse_mask(size, cy, cx, ry, rx):
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32) / size
    return (((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0).astype(np.float32)


def synth_pair(size=512, seed=0):
    """One support/query pair: smooth background field + a bright elliptical 'organ' (slightly moved in the
    query), tiled x3 and z-scored like MR_normalize (dataloaders/dataset_utils.py:101-102,
    ManualAnnoDatasetv2.py:326-327). Returns support [1,3,S,S], fg mask [1,S,S], query [1,3,S,S], query gt."""
    rng = np.random.RandomState(seed + 77)
    out = []
    for j in range(2):
        cy, cx = 0.5 + 0.04 * j, 0.47 + 0.05 * j
        ry, rx = 0.17 - 0.01 * j, 0.22 + 0.01 * j
        m = ellipse_mask(size, cy, cx, ry, rx)
        img = 0.6 * _smooth_field(size, seed + j) + 1.5 * m + 0.05 * rng.randn(size, size).astype(np.float32)
        img = (img - img.mean()) / img.std()
        out.append((torch.from_numpy(np.repeat(img[None, None], 3, axis=1).astype(np.float32)),
                    torch.from_numpy(m[None])))
    (s_img, s_m), (q_img, q_m) = out
    return s_img, s_m, q_img, q_m


def synth_volume(n_slices=32, size=512, seed=0, kind="mri"):
    """[n,S,S] volume with an ellipsoid organ whose cross-section varies with z, + labels [n,S,S]."""
    rng = np.random.RandomState(seed + 991)
    base = _smooth_field(size, seed)
    vol = np.zeros((n_slices, size, size), np.float32)
    lab = np.zeros((n_slices, size, size), np.float32)
    for z in range(n_slices):
        t = (z + 0.5) / n_slices
        r = math.sqrt(max(1e-3, 1.0 - (2 * t - 1) ** 2 * 0.8))
        m = ellipse_mask(size, 0.5 + 0.03 * math.sin(6.0 * t), 0.48 + 0.04 * t, 0.16 * r, 0.21 * r)
        noise = rng.randn(size, size).astype(np.float32)
        if kind == "ct_sparse":
            # a scan whose organ covers only the middle of the z range (the slices outside it have nothing to segment), with
            # five small same-contrast satellites around it (a coarse mask of several components on every organ slice)
            if abs(t - 0.5) > 0.22:
                m = np.zeros_like(m)
                if abs(t - 0.5) > 0.36:          # beyond the body: air only
                    vol[z] = -300.0 + 20.0 * noise
                    continue
            else:
                for j in range(5):
                    a = 2.0 * math.pi * (j / 5.0 + 0.37 * t)
                    m = np.maximum(m, ellipse_mask(size, 0.5 + 0.33 * math.sin(a), 0.5 + 0.33 * math.cos(a),
                                                   0.022 + 0.004 * j, 0.026))
        if kind in ("ct", "ct_sparse"):
            img = -300.0 + 400.0 * base + 350.0 * m + 20.0 * noise
        else:
            img = np.abs(0.6 * base + 1.5 * m + 0.05 * noise + 0.8)
        vol[z], lab[z] = img, m
    vol = (vol - vol.mean()) / vol.std()
    return torch.from_numpy(vol), torch.from_numpy(lab)


MULTI_ORGANS = ((0.30, 0.30, 0.10, 0.13, 1.6), (0.68, 0.70, 0.13, 0.10, -1.4), (0.30, 0.72, 0.09, 0.14, 0.9),
                (0.72, 0.28, 0.11, 0.09, -0.8))


def synth_pair_multi(size=1024, seed=0):
    """A support/query pair with FOUR organs (ellipses of different contrast; slightly moved / scaled in the query) for the
    multi-class configuration (BASELINE config 5): the reference handles classes as an outer loop of 1-way problems over the
    same images (validation.py:207; n_ways == 1 asserted, grid_proto_fewshot.py:172).
    Returns support [1,3,S,S], [4 x fg mask [1,S,S]], query [1,3,S,S], [4 x query gt [1,S,S]]."""
    rng = np.random.RandomState(seed + 177)
    imgs, masks = [], []
    for j in range(2):
        img = 0.4 * _smooth_field(size, seed + j) + 0.05 * rng.randn(size, size).astype(np.float32)
        ms = []
        for (cy, cx, ry, rx, a) in MULTI_ORGANS:
            m = ellipse_mask(size, cy + 0.02 * j, cx - 0.015 * j, ry * (1 - 0.05 * j), rx * (1 + 0.04 * j))
            img = img * (1 - m) + (a + 0.15 * _smooth_field(size, seed + 10 + j, 6)) * m
            ms.append(torch.from_numpy(m[None]))
        img = (img - img.mean()) / img.std()
        imgs.append(torch.from_numpy(np.repeat(img[None, None], 3, axis=1).astype(np.float32)))
        masks.append(ms)
    return imgs[0], masks[0], imgs[1], masks[1]
