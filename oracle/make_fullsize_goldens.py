"""Full-depth oracle taps for BASELINE.json configs 3 / 4 / 5 (test infrastructure; see oracle/__init__.py).

  python oracle/make_fullsize_goldens.py [3] [4] [5] [44]   # writes tests/golden/fullsize_cfg{3,4,5}.npz (44: cfg4_heavytail)
  python oracle/make_fullsize_goldens.py 300 400            # every slice of config 3 / 4: tests/golden/fullvolume_cfg{3,4}.npz
  python oracle/make_fullsize_goldens.py 400 --wseed 777 --vseed 5 --stride 4
      # the same for another weight draw / another synthetic volume (every 4th slice): fullvolume_cfg4_w777_v5.npz
      # (round 5: "within 1e-3" is held on several draws, not on one - tests/test_fullsize_gpu.py VOLUME_VARIANTS)
  python oracle/make_fullsize_goldens.py 4400 --stride 4     # round 6: the same gate on heavy-tailed SAM-H weights: fullvolume_cfg4_heavytail.npz

The CPU oracle (pinned against the reference by oracle/validate_against_reference.py) is run ONCE, here in the build
container, at the configurations' full model depth on seeded synthetic slices; `tests/test_fullsize_gpu.py` runs the HIP
path on the same slices on the GPU box and compares against these records (sigmoid(low_res_masks) within 1e-3, Dice of the
final mask). Running the full-depth oracle inside the GPU tests instead would cost minutes of host time per run.

Inputs come from the seeded generators in protosam_amd/synth.py + protosam_amd/runner.py (`synth_volume`, `support_set`,
`part_assign`: the caller's data contract of validation_protosam.py:346-388), weights from `synth_state_dict(seed 1234)`;
both are reproduced bit for bit on any machine.

  config 3: DINOv2 ViT-B/14 x12 + ALP + SAM ViT-B x12, 512x512 MRI-like volume of 32 slices, default flags + use_cca
  config 4: ... + SAM ViT-H x32, 512x512x64 CT-like volume (the benchmark's workload)
  config 5: DINOv2 ViT-B/14 x12 at 1022^2 (73x73 grid) + MedSAM ViT-B x12, 1024x1024 slice with four organs = four classes
            as four 1-way passes sharing one encoder forward (validation.py:207; n_ways == 1, grid_proto_fewshot.py:172)
Records per case: sigmoid(low_res_masks) of the kept mask token as uint16 fixed point (p * 65535: 8e-6 resolution on the
quantity the 1e-3 bound applies to), final mask (packed bits), scores, coarse foreground probability map every 4th pixel.
"""
import os
import sys
import time

import numpy as np
import torch


def prob16(low):
    """sigmoid(logits) -> uint16 fixed point."""
    return torch.round(torch.sigmoid(low.double()) * 65535.0).numpy().astype(np.uint16)


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")

from protosam_amd.synth_cases import cfg5_inputs, volume_config  # noqa: E402  (the seeded inputs live with the other generators)


def _weights(sam_type, image_size, heavy_tail=False, seed=1234):
    from protosam_amd.grid_proto_fewshot import FewShotSeg
    from protosam_amd.runner import ALP_CFG
    from protosam_amd.segment_anything import sam_model_registry
    from protosam_amd.synth import synth_state_dict
    alp_sd = synth_state_dict(FewShotSeg(image_size, None, dict(ALP_CFG)), seed)
    enc_sd = {k[len("encoder."):]: v for k, v in alp_sd.items() if k.startswith("encoder.")}
    sam_sd = synth_state_dict(sam_model_registry[sam_type](), seed)
    if heavy_tail:
        from protosam_amd.synth import heavy_tail_sam_
        heavy_tail_sam_(sam_sd, seed)
    return enc_sd, sam_sd


def make_volume_config(cfg):
    from oracle import alp as oalp, dinov2 as odino, glue
    from protosam_amd.runner import part_assign, support_set
    from protosam_amd.synth import synth_volume
    sam_type, n, kind, slices, flagsets = volume_config(cfg)
    enc_sd, sam_sd = _weights(sam_type, 512, heavy_tail=(cfg == 44))
    vol, lab = synth_volume(n, 512, seed=0, kind=kind)
    svol, slab = synth_volume(n, 512, seed=1, kind=kind)
    sup_imgs, sup_masks = support_set(svol, slab)
    enc = lambda im: odino.forward_features(im, enc_sd, "dinov2_b14")["x_norm_patchtokens"]  # noqa: E731
    out = {}
    for z in slices:
        t0 = time.time()
        q = vol[z][None, None].repeat(1, 3, 1, 1).contiguous()
        part = part_assign(z, n)
        with torch.no_grad():
            logits = oalp.fewshot_forward(enc, sup_imgs[part], sup_masks[part], q, 512)
            feats = None
            for fname, fl in flagsets.items():
                taps = {}
                pred, scores = glue.protosam_forward(q, logits, sam_sd, sam_type, use_bbox=True, use_points=True,
                                                     point_mode="both", taps=taps, features=feats, **fl)
                feats = taps["features"]
                k = f"z{z}_{fname}"
                out[k + "_mask"] = np.packbits(pred.numpy().astype(bool))
                out[k + "_scores"] = np.array(scores, dtype=np.float32)
                out[k + "_prob"] = prob16(torch.stack([l[0] for l in taps["low_res"]]))
        out[f"z{z}_coarse_p"] = torch.round(logits.double().softmax(1)[0, 1, ::4, ::4] * 65535.0).numpy().astype(np.uint16)
        print(f"config {cfg} z={z}: {len(scores)} component(s), fg {int(pred.sum())} px, {time.time() - t0:.0f}s", flush=True)
    path = os.path.join(GOLD, f"fullsize_cfg{cfg}.npz" if cfg != 44 else "fullsize_cfg4_heavytail.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB)")


AMB_TOL = (4.01e-3, 2.0e-3)


def ambiguous_pixels(lows, size):
    """How many pixels of the FINAL mask an implementation within the north-star tolerance may legitimately flip: a probability
    error <= 1e-3 on sigmoid(low_res) is a logit error <= 4e-3 next to the threshold (dp = p (1 - p) dl), bilinear up-sampling is
    a convex combination, so only pixels whose ORACLE up-sampled logit lies within 4e-3 of zero - for any component of the union -
    can change sign. int32 [2]: the counts at |logit| <= 4.01e-3 and <= 2e-3 after the reference's post-processing (postprocess_masks
    to 1024^2, union, nearest to the slice's size: ProtoSAM.py:669-676)."""
    from oracle import sam_prompt_decoder as odec
    up = odec.postprocess_masks(torch.stack(lows)[None], (1024, 1024), (1024, 1024), "upstream")[0]        # [n, 1024, 1024]
    near = up.abs().min(dim=0).values
    near = torch.nn.functional.interpolate(near[None, None], size=size, mode="nearest")[0, 0]
    return np.array([int((near <= t).sum()) for t in AMB_TOL], dtype=np.int32)


TIE_TOL = 1e-3


def prompt_record(taps):
    """The oracle's DISCRETE decisions of a slice, so that a test can tell a tolerance-sized logit difference from a different
    prompt: float32 [components, 8] = most confident point (x, y), centroid (x, y), box (x0, y0, x1, y1) in the 1024^2 frame
    (ProtoSAM.py:242-289, 349-450), and the packed 1024 x 1024 mask of every pixel whose foreground probability lies within 1e-3
    of its component's maximum: an implementation whose coarse probabilities are within the north-star 1e-3 may pick ANY of those as
    the most confident point (torch.topk over near-ties; softmax saturating to 1.0 inside a confident region makes them many)."""
    pts, boxes = np.asarray(taps["points"], dtype=np.float32), np.asarray(taps["bboxes"], dtype=np.float32)
    rec = np.concatenate([pts.reshape(len(pts), -1)[:, :4], boxes.reshape(len(boxes), 4)], axis=1).astype(np.float32)
    fg = taps["output_p"][0, 1].numpy()
    labels = taps["cc"][1]
    tie = np.zeros(fg.shape, dtype=bool)
    for cid in np.unique(labels):
        if cid == 0:
            continue
        comp = labels == cid
        tie |= comp & (fg >= fg[comp].max() - TIE_TOL)
    return rec, np.packbits(tie)


def volume_record_name(cfg, wseed=1234, vseed=0):
    base = "fullvolume_cfg4_heavytail" if cfg == 44 else f"fullvolume_cfg{cfg}"      # 44: config 4 with synth.heavy_tail_sam_ weights
    return f"{base}.npz" if (wseed, vseed) == (1234, 0) else f"{base}_w{wseed}_v{vseed}.npz"


def make_whole_volume(cfg, wseed=1234, vseed=0, stride=1):
    """EVERY slice of config 3 (32) / config 4 (64), default flags: final mask (packed bits), scores, sigmoid(low_res_masks) of the
    kept token at every 4th pixel (uint16) -> tests/golden/fullvolume_cfg{3,4}.npz. The GPU tests hold BOTH HIP paths (one
    ProtoSAM.forward per slice; 16-slice forward_batch with the LayerNorm folded into the GEMMs) to Dice >= 0.999 against these
    masks, slice by slice (BASELINE.md section 4's gate; validation_protosam.py:169-185).
    `wseed` draws other weights (both models), `vseed` another query volume (the support volume is vseed + 1), `stride` keeps every
    stride-th slice: the variants of tests/test_fullsize_gpu.py VOLUME_VARIANTS."""
    from oracle import alp as oalp, dinov2 as odino, glue
    from protosam_amd.runner import part_assign, support_set
    from protosam_amd.synth import synth_volume
    sam_type, n, kind, _, _ = volume_config(cfg)
    enc_sd, sam_sd = _weights(sam_type, 512, heavy_tail=(cfg == 44), seed=wseed)
    vol, lab = synth_volume(n, 512, seed=vseed, kind=kind)
    svol, slab = synth_volume(n, 512, seed=vseed + 1, kind=kind)
    sup_imgs, sup_masks = support_set(svol, slab)
    path = os.path.join(GOLD, volume_record_name(cfg, wseed, vseed))
    from collections import OrderedDict
    memo = OrderedDict()

    def enc(im):        # the support image of a z-part is encoded once (resize_to_patch_multiple makes a fresh tensor per call: LRU by content)
        key = (float(im.double().sum()), float(im.double().abs().max()), tuple(im.shape))
        if key in memo:
            memo.move_to_end(key)
        else:
            memo[key] = odino.forward_features(im, enc_sd, "dinov2_b14")["x_norm_patchtokens"]
            while len(memo) > 3:
                memo.popitem(last=False)
        return memo[key]
    out = {}
    t_all = time.time()
    zs = list(range(0, n, stride))
    for z in zs:
        t0 = time.time()
        q = vol[z][None, None].repeat(1, 3, 1, 1).contiguous()
        part = part_assign(z, n)
        with torch.no_grad():
            logits = oalp.fewshot_forward(enc, sup_imgs[part], sup_masks[part], q, 512)
            taps = {}
            pred, scores = glue.protosam_forward(q, logits, sam_sd, sam_type, use_bbox=True, use_points=True, point_mode="both",
                                                 use_cca=False, taps=taps)
        out[f"z{z}_mask"] = np.packbits(pred.numpy().astype(bool))
        out[f"z{z}_scores"] = np.array(scores, dtype=np.float32)
        if taps.get("low_res"):
            out[f"z{z}_prob4"] = prob16(torch.stack([l[0] for l in taps["low_res"]]))[..., ::4, ::4].copy()
            out[f"z{z}_amb"] = ambiguous_pixels([l[0] for l in taps["low_res"]], 512)
            out[f"z{z}_prompts"], out[f"z{z}_tie"] = prompt_record(taps)
        print(f"config {cfg} z={z}: {len(scores)} component(s), fg {int(pred.sum())} px, {time.time() - t0:.0f}s "
              f"(total {time.time() - t_all:.0f}s)", flush=True)
        if (z // stride) % 8 == 7 or z == zs[-1]:      # (checkpoint: a long run)
            # (the variants record which slices they hold; the default record keeps its round-4 layout byte for byte)
            extra = {} if (wseed, vseed, stride) == (1234, 0, 1) and cfg != 44 else dict(zs=np.array(zs[:zs.index(z) + 1], dtype=np.int32))
            np.savez_compressed(path, **extra, **out)
    print(f"wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB)")


def make_config5():
    from oracle import alp as oalp, dinov2 as odino, glue
    S = 1024
    enc_sd, sam_sd = _weights("vit_b", S)
    s_img, s_masks, q_img = cfg5_inputs()
    # the four classes share the support / query encoding (one DINOv2 forward at 1022^2 each); resize_to_patch_multiple
    # creates a fresh tensor per call, so the memo is keyed on content
    feats_tok = {}

    def enc_by_content(im):
        key = float(im.double().sum()), tuple(im.shape)
        if key not in feats_tok:
            feats_tok[key] = odino.forward_features(im, enc_sd, "dinov2_b14")["x_norm_patchtokens"]
        return feats_tok[key]
    out = {}
    t0 = time.time()
    for ci, m in enumerate(s_masks):
        with torch.no_grad():
            logits = oalp.fewshot_forward(enc_by_content, s_img, m, q_img, S)
            taps = {}
            seg, conf = glue.protomedsam_forward(q_img, logits, sam_sd, "vit_b", use_cca=True, taps=taps)
        k = f"class{ci}"
        out[k + "_mask"] = np.packbits(seg.numpy().astype(bool))
        out[k + "_coarse_p"] = torch.round(logits.double().softmax(1)[0, 1, ::4, ::4] * 65535.0).numpy().astype(np.uint16)
        if "low" in taps:
            out[k + "_prob"] = prob16(taps["low"][0])
            out[k + "_conf"] = np.asarray(conf[0], dtype=np.float32)
        print(f"config 5 class {ci}: fg {int(seg.sum())} px, {time.time() - t0:.0f}s", flush=True)
    path = os.path.join(GOLD, "fullsize_cfg5.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB)")


if __name__ == "__main__":
    torch.set_num_threads(os.cpu_count() or 1)
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("which", nargs="*", type=int)
    ap.add_argument("--wseed", type=int, default=1234)
    ap.add_argument("--vseed", type=int, default=0)
    ap.add_argument("--stride", type=int, default=1)
    ap.add_argument("--threads", type=int, default=0)
    args = ap.parse_args()
    if args.threads:
        torch.set_num_threads(args.threads)
    which = args.which or [3, 4, 5]
    for c in which:
        if c in (300, 400, 4400):    # every slice of config 3 / 4 (long: ~10 / ~40 minutes on 8 cores); 4400: config 4, heavy-tailed weights
            make_whole_volume(c // 100, args.wseed, args.vseed, args.stride)
        else:
            make_config5() if c == 5 else make_volume_config(c)
