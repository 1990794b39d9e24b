"""Pins the oracle against the real reference (runs ONLY in the build container, where /root/reference exists).

  python oracle/validate_against_reference.py [--write-golden]

1. imports the reference's own modules from /root/reference behind four small shims: stub
   `torchvision` / `cv2` / `kneed`, put `/root/reference/models` on sys.path so `import segment_anything`
   resolves to the vendored copy, make `Tensor.cuda()` the identity, and route `torch.hub.load` to an adapter
   around oracle/dinov2.py (DINOv2 itself is not in the reference tree);
2. runs reference and oracle on the same seeded inputs and asserts agreement (<= 1e-5 unless noted);
3. with --write-golden, stores the REFERENCE outputs (never oracle outputs) as small fixtures under
   tests/golden/, which `tests/test_oracle_golden.py` replays on any machine without the reference.

Nothing here is imported by the product or by the GPU tests.
"""
import argparse
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)


def install_shims():
    import transformers  # noqa: F401  (must be imported before the torchvision stub exists)
    from transformers import Dinov2Config, Dinov2Model  # noqa: F401

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    tv = stub("torchvision")
    tv.models = stub("torchvision.models")
    tv.models.segmentation = stub("torchvision.models.segmentation")
    tv.transforms = stub("torchvision.transforms")
    from PIL import Image  # torchvision's resize(to_pil_image(x), size) IS PIL's bilinear resize
    from oracle import rotate as orot  # torchvision's TENSOR rotate / resize are absent: restated ones are injected
    modes = types.SimpleNamespace(BILINEAR=2, NEAREST=0)

    def tv_resize(im, size, interpolation=modes.BILINEAR, antialias=None):
        if isinstance(im, torch.Tensor):
            return orot.tv_resize(im, size, nearest=(interpolation == modes.NEAREST))
        return im.resize((size[1], size[0]), Image.BILINEAR)

    tv.transforms.functional = stub("torchvision.transforms.functional", resize=tv_resize,
                                    to_pil_image=Image.fromarray,
                                    rotate=lambda im, angle, expand=False: orot.tv_rotate(im, angle, expand=expand),
                                    InterpolationMode=modes)
    tv.ops = stub("torchvision.ops")
    from oracle import amg as oamg  # torchvision.ops NMS is absent: the restated one is injected (oracle/amg.py header)
    tv.ops.boxes = stub("torchvision.ops.boxes", batched_nms=oamg.batched_nms,
                        box_area=lambda b: (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]))   # published contract, XYXY
    stub("cv2")
    stub("kneed")
    sys.path.insert(0, os.path.join(REF, "models"))
    sys.path.insert(0, REF)
    torch.Tensor.cuda = lambda self, *a, **k: self


def close(a, b, tol, what):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    err = (a - b).abs().max().item() if a.numel() else 0.0
    status = "ok" if err <= tol else "FAIL"
    print(f"  [{status}] {what}: max abs diff {err:.3e} (tol {tol:.0e})")
    if err > tol:
        raise SystemExit(f"oracle disagrees with the reference at: {what}")


# ----------------------------------------------------------------------------------------------------------------
def check_alp(gold):
    from models.alpmodule import MultiProtoAsConv  # reference
    from oracle import alp as oalp
    print("ALP module (models/alpmodule.py)")
    from protosam_amd import synth_cases as gi
    qry, sup, msk = gi.alp_case()
    C, hw = qry.shape[2], qry.shape[-1]
    ref_unit = MultiProtoAsConv(proto_grid=[8, 8], feature_hw=[hw, hw], embed_dim=C)
    for mode in ("mask", "gridconv", "gridconv+"):
        ref = ref_unit(qry, sup, msk, mode, 0.95, isval=True, val_wsize=2)[0]
        out, _ = oalp.cls_unit(qry[0], sup[0, 0], msk[0], mode, 0.95, 2)
        close(out, ref, 1e-5, f"mode={mode}")
        gold[f"alp_{mode}"] = ref.numpy()


class _HubAdapter(torch.nn.Module):
    """What `torch.hub.load('facebookresearch/dinov2', ...)` is replaced with: oracle DINOv2 over a seeded state dict."""

    def __init__(self, which, sd, depth):
        super().__init__()
        self.which, self.sd, self.depth = which, sd, depth

    def forward_features(self, x):
        from oracle import dinov2 as odino
        return odino.forward_features(x, self.sd, self.which, depth=self.depth)


def check_fewshot(gold):
    from oracle import alp as oalp, dinov2 as odino
    from protosam_amd import synth_cases as gi
    print("FewShotSeg.forward (models/grid_proto_fewshot.py) with the hub encoder replaced by oracle/dinov2.py")
    depth = gi.FEWSHOT_DEPTH
    enc_sd = gi.fewshot_encoder_sd()
    torch.hub.load = lambda repo, name, **k: _HubAdapter("dinov2_b14", enc_sd, depth)
    from models.grid_proto_fewshot import FewShotSeg  # reference
    for size in gi.FEWSHOT_SIZES:
        cfg = {"which_model": "dinov2_b14", "cls_name": "grid_proto", "proto_grid_size": 8, "lora": 0, "align": False,
               "debug": False, "use_coco_init": False}
        ref_model = FewShotSeg(size, None, cfg).eval()
        s_img, s_m, q_img, _ = gi.fewshot_pair(size)
        with torch.no_grad():
            ref = ref_model([[s_img]], [[s_m]], [[1 - s_m]], [q_img], True, 2)[0]
        enc = lambda im: odino.forward_features(im, enc_sd, "dinov2_b14", depth=depth)["x_norm_patchtokens"]  # noqa
        out = oalp.fewshot_forward(enc, s_img, s_m, q_img, size)
        close(out, ref, 1e-4, f"logits image_size={size} (|logit| <= 20)")
        gold[f"fewshot_logits_{size}"] = ref.numpy().astype(np.float32)


def check_dinov2_vs_transformers():
    """Independent implementation cross-check (the hub code itself is unavailable => 'parity unpinned')."""
    from transformers import Dinov2Config, Dinov2Model
    from oracle import dinov2 as odino
    from protosam_amd.dinov2 import DinoVisionTransformer
    from protosam_amd.synth import synth_state_dict
    print("DINOv2 restatement vs transformers.Dinov2Model (518x518: no pos-embed interpolation; 504 / 1022: interpolated)")
    depth = 2
    sd = synth_state_dict(DinoVisionTransformer("dinov2_vitb14", depth=depth), 7)
    cfg = Dinov2Config(hidden_size=768, num_hidden_layers=depth, num_attention_heads=12, mlp_ratio=4, image_size=518,
                       patch_size=14, layer_norm_eps=1e-6, layerscale_value=1.0, use_swiglu_ffn=False, qkv_bias=True,
                       attention_probs_dropout_prob=0.0, hidden_dropout_prob=0.0, hidden_act="gelu")
    hf = Dinov2Model(cfg).eval()
    m = {}
    m["embeddings.cls_token"] = sd["cls_token"]
    m["embeddings.mask_token"] = sd["mask_token"]
    m["embeddings.position_embeddings"] = sd["pos_embed"]
    m["embeddings.patch_embeddings.projection.weight"] = sd["patch_embed.proj.weight"]
    m["embeddings.patch_embeddings.projection.bias"] = sd["patch_embed.proj.bias"]
    D = 768
    for i in range(depth):
        p, q = f"blocks.{i}.", f"encoder.layer.{i}."
        W, b = sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"]
        for j, nm in enumerate(("query", "key", "value")):
            m[q + f"attention.attention.{nm}.weight"] = W[j * D:(j + 1) * D]
            m[q + f"attention.attention.{nm}.bias"] = b[j * D:(j + 1) * D]
        m[q + "attention.output.dense.weight"] = sd[p + "attn.proj.weight"]
        m[q + "attention.output.dense.bias"] = sd[p + "attn.proj.bias"]
        m[q + "norm1.weight"], m[q + "norm1.bias"] = sd[p + "norm1.weight"], sd[p + "norm1.bias"]
        m[q + "norm2.weight"], m[q + "norm2.bias"] = sd[p + "norm2.weight"], sd[p + "norm2.bias"]
        m[q + "layer_scale1.lambda1"] = sd[p + "ls1.gamma"]
        m[q + "layer_scale2.lambda1"] = sd[p + "ls2.gamma"]
        m[q + "mlp.fc1.weight"], m[q + "mlp.fc1.bias"] = sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]
        m[q + "mlp.fc2.weight"], m[q + "mlp.fc2.bias"] = sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"]
    m["layernorm.weight"], m["layernorm.bias"] = sd["norm.weight"], sd["norm.bias"]
    missing, unexpected = hf.load_state_dict(m, strict=False)
    assert not unexpected, unexpected
    assert all("mask_token" in k for k in missing), missing
    x = torch.randn((1, 3, 518, 518), generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        ref = hf(pixel_values=x).last_hidden_state
    out = odino.forward_features(x, sd, "dinov2_b14", depth=depth)
    close(out["x_norm_patchtokens"], ref[:, 1:], 2e-5, "x_norm_patchtokens")
    close(out["x_norm_clstoken"], ref[:, 0], 2e-5, "x_norm_clstoken")
    # The sizes the configurations actually run at (grid_proto_fewshot.py:88-91 resizes to (S // 14) * 14: 504 for 512 inputs,
    # 1022 for 1024) take the pos-embed interpolation path. transformers resamples with `size=`; the hub code the reference
    # loads passes `scale_factor=(n + 0.1) / 37` (interpolate_offset 0.1), which the oracle restates. Two measurements:
    #   (a) transformers' own interpolation vs the oracle: the difference IS the known size= / scale_factor= delta;
    #   (b) transformers with the ORACLE'S interpolated pos-embed swapped in: everything else (patch embed, cls handling,
    #       token order, blocks, final norm at these sequence lengths) must agree to rounding.
    emb = hf.embeddings
    orig_interp = emb.interpolate_pos_encoding
    for S in (504, 1022):
        xs = torch.randn((1, 3, S, S), generator=torch.Generator().manual_seed(S))
        n = S // 14
        with torch.no_grad():
            o = odino.forward_features(xs, sd, "dinov2_b14", depth=depth)
            ref_own = hf(pixel_values=xs, interpolate_pos_encoding=True).last_hidden_state if "interpolate_pos_encoding" in \
                hf.forward.__code__.co_varnames else hf(pixel_values=xs).last_hidden_state
            pe_oracle = odino.interpolate_pos_encoding(sd["pos_embed"], n, n, False, 0.1)
            emb.interpolate_pos_encoding = lambda embeddings, height, width: pe_oracle
            try:
                ref_swapped = hf(pixel_values=xs).last_hidden_state
            finally:
                emb.interpolate_pos_encoding = orig_interp
            pe_hf = orig_interp(torch.zeros(1, 1 + n * n, 768), S, S)
        d_pe = (pe_hf - pe_oracle).abs().max().item()
        d_own = (o["x_norm_patchtokens"] - ref_own[:, 1:]).abs().max().item()
        print(f"  [info] {S}x{S} ({n}x{n} patches): pos-embed size= vs scale_factor=(n+0.1)/37: max |delta| {d_pe:.3e} on the "
              f"table (std 0.5), {d_own:.3e} on x_norm_patchtokens after {depth} blocks")
        close(o["x_norm_patchtokens"], ref_swapped[:, 1:], 3e-5, f"{S}x{S}: x_norm_patchtokens with the oracle's pos-embed swapped in")
        close(o["x_norm_clstoken"], ref_swapped[:, 0], 3e-5, f"{S}x{S}: x_norm_clstoken with the oracle's pos-embed swapped in")


def _small_encoder_kwargs():
    from functools import partial
    from protosam_amd import synth_cases as gi
    c = gi.SMALL_ENCODER
    return dict(depth=c["depth"], embed_dim=c["embed_dim"], img_size=1024, mlp_ratio=4,
                norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), num_heads=c["num_heads"], patch_size=16,
                qkv_bias=True, use_rel_pos=True, global_attn_indexes=c["global_attn_indexes"], window_size=14,
                out_chans=c["out_chans"])


def check_sam_encoder(gold):
    from segment_anything.modeling import ImageEncoderViT  # vendored reference
    from oracle import sam_image_encoder as oenc
    from protosam_amd.synth import synth_state_dict
    print("SAM ImageEncoderViT (reduced width: 3 blocks [window, global, window], dim 64, 2 heads)")
    ref = ImageEncoderViT(**_small_encoder_kwargs()).eval()
    from protosam_amd import synth_cases as gi
    sd = synth_state_dict(ref, gi.SMALL_ENCODER_SEED)
    ref.load_state_dict(sd)
    x = gi.small_encoder_input()
    with torch.no_grad():
        r = ref(x)
    oenc.VIT_CFGS["tiny_test"] = {k: v for k, v in gi.SMALL_ENCODER.items() if k != "out_chans"}
    o = oenc.image_encoder(x, sd, pre="", model_type="tiny_test")
    close(o, r, 2e-5, "image embedding [1,32,64,64]")
    gold["sam_encoder_small_out"] = r.numpy().astype(np.float32)


def check_sam_decoder(gold):
    from segment_anything import sam_model_registry  # vendored
    from segment_anything.modeling import Sam
    from oracle import sam_prompt_decoder as odec
    from protosam_amd.synth import synth_state_dict
    print("SAM PromptEncoder + MaskDecoder + postprocess (full-size decoder, vendored reference)")
    sam = sam_model_registry["vit_b"]()  # random init, SamBatched
    from protosam_amd import synth_cases as gi
    sd_full = synth_state_dict(sam, gi.DECODER_SEED)
    sd = {k: v for k, v in sd_full.items() if not k.startswith("image_encoder.")}
    sam.load_state_dict(sd, strict=False)
    feats = gi.decoder_features()
    cases = gi.decoder_cases()
    with torch.no_grad():
        pe_ref = sam.prompt_encoder.get_dense_pe()
        close(odec.dense_pe(sd), pe_ref, 1e-5, "get_dense_pe")
        for name, (pc, pl, bx) in cases.items():
            pts = (pc, pl) if pc is not None else None
            sp_r, de_r = sam.prompt_encoder(points=pts, boxes=bx, masks=None)
            sp_o, de_o = odec.prompt_encoder(sd, pts, bx)
            close(sp_o, sp_r, 1e-5, f"{name}: sparse embeddings")
            close(de_o, de_r, 1e-6, f"{name}: dense embeddings")
            for mm in (True, False):
                low_r, iou_r = sam.mask_decoder(image_embeddings=feats, image_pe=pe_ref, sparse_prompt_embeddings=sp_r,
                                                dense_prompt_embeddings=de_r, multimask_output=mm)
                low_o, iou_o = odec.mask_decoder(sd, feats, odec.dense_pe(sd), sp_o, de_o, mm)
                close(low_o, low_r, 2e-4, f"{name}: low_res_masks multimask={mm}")
                close(iou_o, iou_r, 2e-5, f"{name}: iou_predictions multimask={mm}")
                if mm:
                    gold[f"dec_{name}_low_res"] = low_r.numpy().astype(np.float32)
                    gold[f"dec_{name}_iou"] = iou_r.numpy().astype(np.float32)
        low = low_r
        # mask prompts (prompt_encoder.py:102-105,163-164): two blob masks with the 10 / 248 values ProtoSAM hands over
        mk = gi.mask_prompt_case()
        sp_r, de_r = sam.prompt_encoder(points=None, boxes=None, masks=mk)
        sp_o, de_o = odec.prompt_encoder(sd, None, None, masks=mk)
        assert sp_r.shape == sp_o.shape == (2, 0, 256)
        close(de_o, de_r, 1e-4, "mask prompt: dense embeddings (values up to ~1e2)")
        lm_r, im_r = sam.mask_decoder(image_embeddings=feats, image_pe=pe_ref, sparse_prompt_embeddings=sp_r,
                                      dense_prompt_embeddings=de_r, multimask_output=True)
        lm_o, im_o = odec.mask_decoder(sd, feats, odec.dense_pe(sd), sp_o, de_o, True)
        close(lm_o, lm_r, 5e-4, "mask prompt: low_res_masks")
        close(im_o, im_r, 5e-5, "mask prompt: iou_predictions")
        gold["dec_mask_dense"] = de_r[:, :, ::8, ::8].numpy().astype(np.float32)
        gold["dec_mask_low_res"] = lm_r.numpy().astype(np.float32)
        gold["dec_mask_iou"] = im_r.numpy().astype(np.float32)
        r_b = sam.postprocess_masks(low, (1024, 1024), (1024, 1024))           # SamBatched: align_corners=True
        close(odec.postprocess_masks(low, (1024, 1024), (1024, 1024), "batched"), r_b, 1e-5, "postprocess SamBatched")
        r_n = Sam.postprocess_masks(sam, low, (1024, 1024), (512, 512))        # vendored Sam: nearest
        close(odec.postprocess_masks(low, (1024, 1024), (512, 512), "nearest"), r_n, 0, "postprocess Sam(nearest)")
        gold["post_batched_row"] = r_b[0, 0, 511].numpy().astype(np.float32)
    # ResizeLongestSide coordinate maps
    from segment_anything.utils.transforms import ResizeLongestSide
    tr = ResizeLongestSide(1024)
    pts = np.array([[10.0, 20.0], [511.0, 300.5]])
    close(odec.apply_coords(pts, (512, 512)), tr.apply_coords(pts, (512, 512)), 0, "ResizeLongestSide.apply_coords")
    close(odec.apply_coords(pts, (600, 900)), tr.apply_coords(pts, (600, 900)), 0, "apply_coords non-square")


def check_glue(gold):
    from oracle import glue
    import scipy.ndimage as ndi
    print("connected components / prompt extraction (cv2 absent: cross-check against scipy.ndimage)")
    rng = np.random.RandomState(5)
    for trial in range(4):
        img = (ndi.gaussian_filter(rng.randn(96, 128), 3 + trial) > 0.02).astype(np.uint8)
        n, labels, stats, cent = glue.connected_components_with_stats(img)
        lab2, n2 = ndi.label(img, structure=np.ones((3, 3)))
        assert n == n2 + 1, (n, n2)
        # same partition (label numbering may differ): map through first pixel
        for j in range(1, n):
            ys, xs = np.nonzero(labels == j)
            l2 = lab2[ys[0], xs[0]]
            assert np.array_equal(labels == j, lab2 == l2)
            com = ndi.center_of_mass(img, lab2, l2)
            close(cent[j], np.array([com[1], com[0]]), 1e-9, f"trial {trial} centroid {j}") if j == 1 else None
            assert stats[j, 4] == (lab2 == l2).sum()
    # reference util functions that do not need cv2
    from util.utils import get_confidence_from_logits
    lg = torch.randn((1, 2, 40, 40), generator=torch.Generator().manual_seed(2)) * 3
    close(glue.confidence_from_logits(lg), get_confidence_from_logits(lg), 1e-6, "get_confidence_from_logits")


def _amg_reference_model():
    """Vendored registry's SamBatched vit_b with the encoder's block stack truncated, synthetic weights."""
    from functools import partial
    from segment_anything import sam_model_registry
    from segment_anything.modeling import ImageEncoderViT
    from protosam_amd import synth_cases as gi
    from protosam_amd.synth import synth_state_dict
    sam = sam_model_registry["vit_b"]()
    sam.image_encoder = ImageEncoderViT(depth=gi.AMG_ENCODER_DEPTH, embed_dim=768, img_size=1024, mlp_ratio=4,
                                        norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), num_heads=12, patch_size=16,
                                        qkv_bias=True, use_rel_pos=True, global_attn_indexes=[2, 5, 8, 11],
                                        window_size=14, out_chans=256)
    sd = synth_state_dict(sam, gi.AMG_SEED)
    sam.load_state_dict(sd)
    return sam.eval(), sd


def amg_thresholds(iou_all, stab_all):
    """Synthetic weights do not produce calibrated scores, so the generator's two thresholds are placed inside the widest
    gap of a central quantile band of each score: a good share of the candidates passes each filter and none sits on the
    boundary (the predicted-IoU one must be > 0 to be applied at all, automatic_mask_generator.py:295)."""
    def gap_mid(v, qlo, qhi):
        v = np.sort(np.asarray(v, dtype=np.float64)[np.isfinite(v)])
        lo, hi = int(qlo * len(v)), int(qhi * len(v))
        k = lo + int(np.argmax(np.diff(v[lo:hi + 1])))
        return float((v[k] + v[k + 1]) / 2)
    t_iou = gap_mid(iou_all, 0.5, 0.7)
    t_stab = gap_mid(np.asarray(stab_all)[np.asarray(iou_all) > t_iou], 0.3, 0.6)   # the filters apply in sequence
    assert t_iou > 0 and t_stab > 0
    return t_iou, t_stab


def check_amg(gold):
    from segment_anything import SamAutomaticMaskGenerator  # vendored
    from oracle import amg as oamg
    from protosam_amd import synth_cases as gi
    print("SamAutomaticMaskGenerator.generate + SamWrapper.forward (vendored reference; NMS = injected restatement)")
    sam, sd = _amg_reference_model()
    img, label = gi.amg_case()
    taps = {}
    oamg.generate(img, sd, encoder_depth=gi.AMG_ENCODER_DEPTH, pred_iou_thresh=0.0, stability_score_thresh=0.0,
                  taps=taps, **gi.AMG_ARGS)
    t_iou, t_stab = amg_thresholds(taps["iou_all"].numpy(), taps["stab_all"].numpy())
    print(f"  thresholds: pred_iou {t_iou:.6f}, stability {t_stab:.6f}")
    kw = dict(pred_iou_thresh=t_iou, stability_score_thresh=t_stab, **gi.AMG_ARGS)
    with torch.no_grad():
        ref = SamAutomaticMaskGenerator(sam, **kw).generate(img)
    ora = oamg.generate(img, sd, encoder_depth=gi.AMG_ENCODER_DEPTH, features=taps["features"], **kw)
    assert len(ref) == len(ora) and len(ref) > 3, (len(ref), len(ora))
    print(f"  {len(ref)} masks kept of {len(taps['iou_all'])} candidates")
    for k in ("predicted_iou", "stability_score"):
        close([r[k] for r in ora], [r[k] for r in ref], 2e-5, k)
    for k in ("bbox", "area", "point_coords", "crop_box"):
        close(np.array([r[k] for r in ora], dtype=np.float64).reshape(len(ref), -1),
              np.array([r[k] for r in ref], dtype=np.float64).reshape(len(ref), -1), 0, k)
    diff = max(int((a["segmentation"] != b["segmentation"]).sum()) for a, b in zip(ora, ref))
    print(f"  [{'ok' if diff <= 4 else 'FAIL'}] segmentation: at most {diff} differing pixels per mask")
    assert diff <= 4
    # synthetic weights give noise-like masks whose boxes all span the image, so AMG_ARGS disables suppression
    # (box_nms_thresh = 1.0) to compare many records; with the default 0.7 exactly the top-scoring one survives
    kw7 = dict(kw, box_nms_thresh=0.7)
    with torch.no_grad():
        ref7 = SamAutomaticMaskGenerator(sam, **kw7).generate(img)
    ora7 = oamg.generate(img, sd, encoder_depth=gi.AMG_ENCODER_DEPTH, features=taps["features"], **kw7)
    assert len(ref7) == len(ora7) == 1 and ref7[0]["point_coords"] == ora7[0]["point_coords"] == ref[0]["point_coords"]
    print("  [ok] box_nms_thresh=0.7: the single survivor is the top-scoring record")
    # SamWrapper.forward on top of it
    import models.SamWrapper as ref_sw
    w = ref_sw.SamWrapper.__new__(ref_sw.SamWrapper)
    torch.nn.Module.__init__(w)
    w.sam = sam
    w.mask_generator = SamAutomaticMaskGenerator(sam, **kw)
    from segment_anything.utils.transforms import ResizeLongestSide
    w.transform = ResizeLongestSide(1024)
    with torch.no_grad():
        best_ref = w(img, label)
    best_ora, bi, ious, _ = oamg.sam_wrapper_forward(img, label, sd, encoder_depth=gi.AMG_ENCODER_DEPTH,
                                                     features=taps["features"], **kw)
    d = int((best_ref != best_ora).sum())
    print(f"  [{'ok' if d <= 4 else 'FAIL'}] SamWrapper.forward: best mask #{bi} (IoU {float(ious[bi]):.4f}), {d} differing pixels")
    assert d <= 4
    # crop layers (crop_n_layers = 1: the image + 2 x 2 overlapping crops, re-encoded each) and small-region removal
    # (min_mask_region_area > 0; cv2.connectedComponentsWithStats comes from the oracle's restatement, as everywhere else) on
    # a SMALL image (gi.amg_small_case: see there why). Two settings: no suppression (every candidate of every crop becomes a
    # record and goes through the hole / island removal) and the default crop_nms_thresh (one survivor per crop: the
    # cross-crop NMS that prefers smaller crops).
    _install_cv2_restatements()
    img_s = gi.amg_small_case()
    hs, ws_ = img_s.shape[:2]
    for tag, extra in (("all", dict(crop_nms_thresh=1.0)), ("nms", dict())):
        kwc = dict(gi.AMG_CROP_ARGS, **extra)
        with torch.no_grad():
            ref_c = SamAutomaticMaskGenerator(sam, **kwc).generate(img_s)
        ora_c = oamg.generate(img_s, sd, encoder_depth=gi.AMG_ENCODER_DEPTH, **kwc)
        n_crop = sum(1 for r in ref_c if r["crop_box"] != [0, 0, ws_, hs])
        print(f"  crops + small regions ({tag}): {len(ref_c)} records, {n_crop} from the four layer-1 crops")
        assert len(ref_c) == len(ora_c) and n_crop > 0 and len(ref_c) > n_crop
        for k in ("predicted_iou", "stability_score"):
            close([r[k] for r in ora_c], [r[k] for r in ref_c], 2e-5, f"crops {tag}: {k}")
        for k in ("bbox", "area", "point_coords", "crop_box"):
            close(np.array([r[k] for r in ora_c], dtype=np.float64).reshape(len(ref_c), -1),
                  np.array([r[k] for r in ref_c], dtype=np.float64).reshape(len(ref_c), -1), 0, f"crops {tag}: {k}")
        diff = max(int((a["segmentation"] != b["segmentation"]).sum()) for a, b in zip(ora_c, ref_c))
        print(f"  [{'ok' if diff == 0 else 'FAIL'}] crops {tag}: segmentation: at most {diff} differing pixels per mask")
        assert diff == 0
        if tag == "all":
            plain = SamAutomaticMaskGenerator(sam, **dict(kwc, min_mask_region_area=0)).generate(img_s)
            changed = sum(1 for a, b in zip(ref_c, plain) if a["area"] != b["area"])
            print(f"  (hole / island removal changed {changed} of {len(ref_c)} masks)")
            assert changed > 10
        gold[f"amgc_{tag}_pred_iou"] = np.array([r["predicted_iou"] for r in ref_c], dtype=np.float32)
        gold[f"amgc_{tag}_stability"] = np.array([r["stability_score"] for r in ref_c], dtype=np.float32)
        gold[f"amgc_{tag}_bbox"] = np.array([r["bbox"] for r in ref_c], dtype=np.int32)
        gold[f"amgc_{tag}_area"] = np.array([r["area"] for r in ref_c], dtype=np.int32)
        gold[f"amgc_{tag}_points"] = np.array([r["point_coords"][0] for r in ref_c], dtype=np.float64)
        gold[f"amgc_{tag}_crop_box"] = np.array([r["crop_box"] for r in ref_c], dtype=np.int32)
        gold[f"amgc_{tag}_mask_bits"] = np.stack([np.packbits(r["segmentation"]) for r in ref_c])
    gold["amg_thresholds"] = np.array([t_iou, t_stab], dtype=np.float64)
    gold["amg_iou_all"] = taps["iou_all"].numpy().astype(np.float32)   # oracle values; the reference exposes kept ones only
    gold["amg_pred_iou"] = np.array([r["predicted_iou"] for r in ref], dtype=np.float32)
    gold["amg_stability"] = np.array([r["stability_score"] for r in ref], dtype=np.float32)
    gold["amg_bbox"] = np.array([r["bbox"] for r in ref], dtype=np.int32)
    gold["amg_area"] = np.array([r["area"] for r in ref], dtype=np.int32)
    gold["amg_points"] = np.array([r["point_coords"][0] for r in ref], dtype=np.float64)
    gold["amg_best_mask_bits"] = np.packbits(best_ref)
    gold["amg_best_index"] = np.array([bi], dtype=np.int32)


def check_rotate(gold):
    """util/utils.py `rotate_tensor_no_crop` / `reverse_tensor` (the reference's own code: canvas bookkeeping, which
    interpolation for which channel count, resize-then-rotate-then-crop order) against oracle/rotate.py. The torchvision
    primitives underneath are the restated ones in both (torchvision is absent), so this pins the helpers, not `rotate` /
    `resize` themselves."""
    import matplotlib
    matplotlib.use("Agg")
    for name in ("tqdm", "tqdm.auto"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                m = types.ModuleType(name)
                m.tqdm = lambda x, *a, **k: x
                sys.modules[name] = m
    from util import utils as rutils  # reference
    from oracle import rotate as orot
    print("rotation helpers (util/utils.py:40-83)")
    g = torch.Generator().manual_seed(77)
    for shape, deg in (((1, 3, 96, 96), 15), ((2, 3, 64, 80), -30), ((1, 1, 64, 64), 20), ((1, 3, 50, 50), 0)):
        x = torch.randn(shape, generator=g)
        ref, ref_sz = rutils.rotate_tensor_no_crop(x, deg)
        got, got_sz = orot.rotate_tensor_no_crop(x, deg)
        assert tuple(ref_sz) == tuple(got_sz), (ref_sz, got_sz)
        close(got, ref, 0.0, f"rotate_tensor_no_crop {shape} by {deg}")
        if deg != 0:
            logits = torch.randn((shape[0], 2) + shape[2:], generator=g)
            close(orot.reverse_tensor(logits, got_sz[0], got_sz[1], -deg),
                  rutils.reverse_tensor(logits, ref_sz[0], ref_sz[1], -deg), 0.0, f"reverse_tensor {shape} by {-deg}")
    # golden vector (reference helper outputs on a small case) for tests/test_oracle_golden.py
    x = torch.randn((1, 3, 48, 48), generator=g)
    lg = torch.randn((1, 2, 48, 48), generator=g)
    r, (rh, rw) = rutils.rotate_tensor_no_crop(x, 15)
    gold["rotate_in"], gold["rotate_logits"] = x.numpy(), lg.numpy()
    gold["rotate_out"], gold["rotate_size"] = r.numpy(), np.array([rh, rw], dtype=np.int32)
    gold["rotate_back"] = rutils.reverse_tensor(lg, rh, rw, -15).numpy()


def _install_cv2_restatements():
    """cv2 is absent: give the stub module the three functions the hot path calls, from the oracle's restatements (the
    same move as `batched_nms` above). What this pins is the REFERENCE'S OWN orchestration around them - prompt assembly,
    label conventions, per-component loops, selection rules - not cv2 itself (stated as unpinned in oracle/glue.py)."""
    from oracle import glue
    cv2 = sys.modules["cv2"]
    cv2.INTER_NEAREST, cv2.COLOR_BGR2RGB = 0, 4

    def connectedComponentsWithStats(img, connectivity=8):
        assert connectivity == 8
        return glue.connected_components_with_stats(img)

    def dilate(img, kernel, iterations=1):
        assert kernel.shape == (3, 3) and kernel.all()
        return glue.dilate3x3(img, iterations)

    def resize(img, dsize, interpolation=None):
        assert interpolation == cv2.INTER_NEAREST
        h, w = img.shape[:2]
        ys = np.minimum(np.floor(np.arange(dsize[1]) * (h / dsize[1])).astype(np.int64), h - 1)
        xs = np.minimum(np.floor(np.arange(dsize[0]) * (w / dsize[0])).astype(np.int64), w - 1)
        return img[ys][:, xs]

    cv2.connectedComponentsWithStats, cv2.dilate, cv2.resize = connectedComponentsWithStats, dilate, resize


class _FixedCoarse:
    """A ModelWrapper stand-in whose logits are given: the reference only calls `model(input)` (ProtoSAM.py:548)."""

    def __init__(self, logits):
        self.logits = logits

    def __call__(self, inp):
        return self.logits.clone()


def _truncate_vendored_registry(depth):
    """The vendored registry builds the full ViT-B (12 blocks); the orchestration cases use its first `depth` blocks."""
    import segment_anything  # noqa: F401
    bs = sys.modules["segment_anything.build_sam"]   # (the package re-exports a FUNCTION of the same name)
    if not hasattr(bs, "_orig_build_sam"):
        bs._orig_build_sam = bs._build_sam
    def small(encoder_embed_dim, encoder_depth, encoder_num_heads, encoder_global_attn_indexes, checkpoint=None):
        return bs._orig_build_sam(encoder_embed_dim, depth, encoder_num_heads,
                                  [i for i in encoder_global_attn_indexes if i < depth], checkpoint)
    bs._build_sam = small


def _pack(mask):
    return np.packbits(np.asarray(mask).astype(bool))


def check_orchestration(gold, tmpdir):
    """The reference's own ProtoSAM.forward (models/ProtoSAM.py:536-678, incl. :242-289, :349-466, :468-533), SamPredictor
    (models/segment_anything/predictor.py:34-241), ProtoMedSAM.forward (models/ProtoMedSAM.py:122-222), cca /
    get_connected_components (util/utils.py:474-541) executed end to end on CPU, against oracle/glue.py + odec.predict."""
    import matplotlib
    matplotlib.use("Agg")
    _install_cv2_restatements()
    from oracle import alp as oalp, dinov2 as odino, glue
    from protosam_amd import synth_cases as gi
    from oracle import sam_image_encoder as oenc, sam_prompt_decoder as odec
    from protosam_amd.synth import synth_state_dict
    _truncate_vendored_registry(gi.ORCH_SAM_DEPTH)
    import models.ProtoSAM as ref_ps
    import models.ProtoMedSAM as ref_pm
    from segment_anything import SamPredictor, sam_model_registry
    print("orchestration: reference ProtoSAM.forward / SamPredictor / ProtoMedSAM.forward vs oracle/glue.py")
    sam0 = sam_model_registry["vit_b"]()
    sam_sd = synth_state_dict(sam0, gi.ORCH_SAM_SEED)
    ckpt = os.path.join(tmpdir, "sam_vit_b_synth.pth")
    torch.save(sam_sd, ckpt)
    D = gi.ORCH_SAM_DEPTH
    q = gi.orch_query()
    S = gi.ORCH_SIZE

    # -- SamPredictor.set_image / predict ---------------------------------------------------------------------------------
    sam0.load_state_dict(sam_sd)
    predictor = SamPredictor(sam0.eval())
    feats_by_hw = {}
    for name, hw, pc, pl, box, with_mask, mm, rl in gi.predictor_cases():
        img = gi.predictor_image(hw)
        with torch.no_grad():
            predictor.set_image(img)
            mk = gi.mask_prompt_case()[0].numpy() if with_mask else None
            m_r, s_r, low_r = predictor.predict(point_coords=pc, point_labels=pl, box=box, mask_input=mk,
                                                multimask_output=mm, return_logits=rl)
            if hw not in feats_by_hw:
                rz = glue.apply_image(img)
                assert tuple(rz.shape[:2]) == tuple(predictor.input_size)
                feats_by_hw[hw] = (oenc.image_encoder(glue.sam_preprocess(rz), sam_sd, model_type="vit_b", depth=D),
                                   tuple(rz.shape[:2]))
                close(feats_by_hw[hw][0], predictor.get_image_embedding(), 2e-5, f"predictor {hw}: image embedding")
            feats, in_size = feats_by_hw[hw]
            m_o, s_o, low_o = odec.predict(sam_sd, feats, pc, pl, box, mm, hw, variant="batched", mask_input=mk,
                                           input_size=in_size)
        close(low_o, low_r, 5e-4, f"predictor {name}: low_res logits")
        close(s_o, s_r, 2e-5, f"predictor {name}: iou predictions")
        if rl:   # return_logits=True hands back the post-processed logits instead of the thresholded masks
            m_full = odec.postprocess_masks(low_o[None], in_size, hw, "batched")[0]
            close(m_full, m_r, 5e-3, f"predictor {name}: full-size logits")
        else:
            d = int((m_o.numpy() != m_r).sum())
            print(f"  [{'ok' if d <= 4 else 'FAIL'}] predictor {name}: {d} differing mask pixels of {m_r.size}")
            assert d <= 4
        gold[f"pred_{name}_low"] = low_r[..., ::2, ::2].astype(np.float32)     # fp32: the 1e-3 bound is asserted without slack
        gold[f"pred_{name}_iou"] = s_r.astype(np.float32)
    with torch.no_grad():
        predictor.reset_image()
    try:
        predictor.predict(point_coords=np.array([[1.0, 2.0]]), point_labels=np.array([1]))
        raise SystemExit("predict before set_image must raise")
    except RuntimeError:
        print("  [ok] predict before set_image raises RuntimeError")

    # -- ProtoSAM.forward, every flag set, on given coarse logits --------------------------------------------------------
    logits = gi.orch_coarse_logits()
    qf = F_interp(q)
    feats = oenc.image_encoder(glue.sam_preprocess(glue.quantise_image(qf)), sam_sd, model_type="vit_b", depth=D)
    # Shim 5 (scoped to this check): the reference runs with CUDA tensors (validation_protosam.py:303,367), where
    # `x.detach().cpu()` is a COPY. On CPU tensors it aliases, and ProtoSAM.py:361-362 (`bg_p = output_p[0, 0].detach().cpu();
    # bg_p[bg_p < 0.95] = 0`) would then zero `output_p` itself in place, so that the ring search at :409-410 runs over zeros.
    # `.cpu()` is made a copy here so that the CPU run has the device semantics the oracle (and the HIP path) follow.
    orig_cpu = torch.Tensor.cpu
    torch.Tensor.cpu = lambda self, *a, **k: orig_cpu(self, *a, **k).clone()
    try:
        _protosam_flag_cases(gold, ref_ps, gi, glue, q, S, D, logits, sam_sd, ckpt, feats)
    finally:
        torch.Tensor.cpu = orig_cpu
    inp = ref_ps.InputFactory.create_input(ref_ps.TYPE_ALPNET, q, support_images=[q], support_labels=[torch.zeros(1, S, S)],
                                           isval=True, val_wsize=2)
    _orchestration_rest(gold, tmpdir, ref_ps, ref_pm, gi, glue, oalp, odino, q, S, D, logits, sam_sd, ckpt, feats, inp)


def _protosam_flag_cases(gold, ref_ps, gi, glue, q, S, D, logits, sam_sd, ckpt, feats):
    for name, kw in gi.ORCH_FLAGS.items():
        ref_model = ref_ps.ProtoSAM(image_size=(1024, 1024), coarse_segmentation_model=_FixedCoarse(logits),
                                    sam_pretrained_path=ckpt, num_points_for_sam=1, use_sam_trans=True, **kw).eval()
        inp = ref_ps.InputFactory.create_input(ref_ps.TYPE_ALPNET, q, support_images=[q], support_labels=[torch.zeros(1, S, S)],
                                               isval=True, val_wsize=2)
        with torch.no_grad():
            pred_r, scores_r = ref_model(q, inp, degrees_rotate=0)
            taps = {}
            pred_o, scores_o = glue.protosam_forward(q, logits, sam_sd, "vit_b", postprocess="batched", encoder_depth=D,
                                                     features=feats, taps=taps,
                                                     **{k: v for k, v in kw.items()})
        assert pred_r.shape == pred_o.shape == (S, S) and pred_r.dtype == torch.float32
        d = int((pred_r != pred_o).sum())
        print(f"  [{'ok' if d == 0 else 'FAIL'}] ProtoSAM.forward {name}: {len(scores_r)} prompt sets, {d} differing pixels, "
              f"fg {int(pred_r.sum())}")
        assert d == 0 and len(scores_r) == len(scores_o)
        close(np.array(scores_o, dtype=np.float64), np.array([float(v) for v in scores_r]), 1e-5, f"ProtoSAM.forward {name}: scores")
        gold[f"orch_{name}_mask"] = _pack(pred_r.numpy())
        gold[f"orch_{name}_scores"] = np.array([float(v) for v in scores_r], dtype=np.float32)
        # the oracle's low-res logits (bit-equal to the reference's by the mask / score checks above), every 4th pixel
        gold[f"orch_{name}_low"] = torch.stack([torch.as_tensor(l) for l in taps["low_res"]])[..., ::4, ::4].numpy().astype(np.float32)


def _orchestration_rest(gold, tmpdir, ref_ps, ref_pm, gi, glue, oalp, odino, q, S, D, logits, sam_sd, ckpt, feats, inp):
    # empty coarse mask: the 1024 x 1024 arg-max map, un-resized, and [0] (ProtoSAM.py:612-613)
    ref_model = ref_ps.ProtoSAM(image_size=(1024, 1024), coarse_segmentation_model=_FixedCoarse(gi.orch_empty_logits()),
                                sam_pretrained_path=ckpt, use_bbox=True, use_points=True, point_mode="both").eval()
    with torch.no_grad():
        pred_r, scores_r = ref_model(q, inp)
        pred_o, scores_o = glue.protosam_forward(q, gi.orch_empty_logits(), sam_sd, "vit_b", features=feats)
    assert tuple(pred_r.shape) == tuple(pred_o.shape) == (1024, 1024) and int(pred_r.sum()) == 0 and scores_r == scores_o == [0]
    print("  [ok] ProtoSAM.forward with an empty coarse mask: [1024,1024] zeros, [0]")
    # coarse_pred_only (ProtoSAM.py:580-590)
    for use_cca in (False, True):
        ref_model = ref_ps.ProtoSAM(image_size=(1024, 1024), coarse_segmentation_model=_FixedCoarse(logits),
                                    sam_pretrained_path=ckpt, use_bbox=True, use_points=True, coarse_pred_only=True,
                                    use_cca=use_cca).eval()
        with torch.no_grad():
            pred_r, conf_r = ref_model(q, inp)
        pred_o, conf_o = glue.coarse_pred_only(logits, S, use_cca)
        assert torch.equal(torch.as_tensor(pred_r).long(), torch.as_tensor(pred_o).long())
        close(conf_o[0], conf_r[0], 1e-6, f"coarse_pred_only use_cca={use_cca}: confidence")
        gold[f"orch_coarse_only_{int(use_cca)}"] = np.array([float(conf_r[0]), float(torch.as_tensor(pred_r).sum())])
    # constructor errors (ProtoSAM.py:197,200-201)
    for bad, exc in ((dict(use_points=False, use_bbox=False, use_mask=False), AssertionError),
                     (dict(point_mode="nearest"), ValueError)):
        try:
            ref_ps.ProtoSAM((1024, 1024), None, ckpt, **bad)
            raise SystemExit("constructor must raise")
        except exc:
            pass

    # -- the real coarse model in the loop: reference ALPNetWrapper(FewShotSeg) -> ProtoSAM.forward ----------------------
    depth = gi.FEWSHOT_DEPTH
    enc_sd = gi.fewshot_encoder_sd()
    torch.hub.load = lambda repo, name, **k: _HubAdapter("dinov2_b14", enc_sd, depth)
    from models.grid_proto_fewshot import FewShotSeg
    cfg = {"which_model": "dinov2_b14", "cls_name": "grid_proto", "proto_grid_size": 8, "lora": 0, "align": False,
           "debug": False, "use_coco_init": False}
    s_img, s_m, q_img, _ = gi.fewshot_pair(S)
    ref_alp = ref_ps.ALPNetWrapper(FewShotSeg(S, None, cfg).eval())
    ref_model = ref_ps.ProtoSAM(image_size=(1024, 1024), coarse_segmentation_model=ref_alp, sam_pretrained_path=ckpt,
                                num_points_for_sam=1, use_sam_trans=True, **gi.ORCH_FLAGS["default"]).eval()
    inp = ref_ps.InputFactory.create_input(ref_ps.TYPE_ALPNET, q_img, support_images=[s_img], support_labels=[s_m],
                                           isval=True, val_wsize=2)
    enc = lambda im: odino.forward_features(im, enc_sd, "dinov2_b14", depth=depth)["x_norm_patchtokens"]  # noqa: E731
    with torch.no_grad():
        pred_r, scores_r = ref_model(q_img, inp, degrees_rotate=0)
        lg_o = oalp.fewshot_forward(enc, s_img, s_m, q_img, S)
        pred_o, scores_o = glue.protosam_forward(q_img, lg_o, sam_sd, "vit_b", postprocess="batched", encoder_depth=D,
                                                 features=feats, **gi.ORCH_FLAGS["default"])
    d = int((pred_r != pred_o).sum())
    print(f"  [{'ok' if d <= 2 else 'FAIL'}] FewShotSeg -> ProtoSAM.forward: {len(scores_r)} prompt sets, {d} differing pixels")
    assert d <= 2 and len(scores_r) == len(scores_o)
    close(np.array(scores_o), np.array([float(v) for v in scores_r]), 2e-5, "FewShotSeg -> ProtoSAM.forward: scores")
    gold["orch_alp_mask"] = _pack(pred_r.numpy())
    gold["orch_alp_scores"] = np.array([float(v) for v in scores_r], dtype=np.float32)

    # -- ProtoMedSAM.forward (box prompts, [0,1] hand-off, sigmoid before the resize) ------------------------------------
    ckpt_med = os.path.join(tmpdir, "medsam_vit_b_synth.pth")
    torch.save(sam_sd, ckpt_med)
    for use_cca in (True,):
        ref_med = ref_pm.ProtoMedSAM((1024, 1024), _FixedCoarse(logits), ckpt_med, use_cca=use_cca).eval()
        with torch.no_grad():
            seg_r, conf_r = ref_med(q, inp)
            seg_o, conf_o = glue.protomedsam_forward(q, logits, sam_sd, "vit_b", use_cca=use_cca, encoder_depth=D)
        d = int((torch.as_tensor(seg_r).long() != seg_o.long()).sum())
        print(f"  [{'ok' if d <= 2 else 'FAIL'}] ProtoMedSAM.forward use_cca={use_cca}: {d} differing pixels, fg {int(seg_r.sum())}")
        assert d <= 2 and tuple(seg_r.shape) == (S, S)
        close(np.asarray(conf_o[0]), np.asarray(conf_r[0]), 2e-5, "ProtoMedSAM.forward: confidence")
        gold["orch_medsam_mask"] = _pack(torch.as_tensor(seg_r).numpy())
        gold["orch_medsam_conf"] = np.asarray(conf_r[0], dtype=np.float32)
    with torch.no_grad():
        seg_r, conf_r = ref_pm.ProtoMedSAM((1024, 1024), _FixedCoarse(gi.orch_empty_logits()), ckpt_med, use_cca=True).eval()(q, inp)
        seg_o, conf_o = glue.protomedsam_forward(q, gi.orch_empty_logits(), sam_sd, "vit_b", use_cca=True, encoder_depth=D)
    assert tuple(seg_r.shape) == tuple(seg_o.shape) == (S, S) and int(seg_r.sum()) == 0 and conf_r == conf_o == [0]
    print("  [ok] ProtoMedSAM.forward with an empty coarse mask: [512,512] zeros, [0]")

    # -- the caller's metric (validation_protosam.py:169-185) -------------------------------------------------------------
    check_metric(gold)


def F_interp(q):
    return torch.nn.functional.interpolate(q, size=(1024, 1024), mode="bilinear")


def check_metric(gold):
    """`get_dice_iou_precision_recall` lives in validation_protosam.py, which imports sacred at module level (absent): the
    function's source lines are executed from the file itself, nothing else of the module is."""
    import ast
    from protosam_amd.metrics import get_dice_iou_precision_recall as ours
    src = open(os.path.join(REF, "validation_protosam.py")).read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "get_dice_iou_precision_recall"][0]
    ns = {"torch": torch}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "validation_protosam.py", "exec"), ns)
    ref = ns["get_dice_iou_precision_recall"]
    g = torch.Generator().manual_seed(9)
    a = (torch.rand((64, 64), generator=g) > 0.6).float()
    b = (torch.rand((64, 64), generator=g) > 0.5).float()
    r, o = ref(a, b), ours(a, b)
    assert set(r) == set(o)
    for k in r:
        close(o[k], r[k], 0, f"get_dice_iou_precision_recall: {k}")
    z = ref(a, torch.zeros_like(b))
    assert z == ours(a, torch.zeros_like(b)) == {"dice": 0, "precision": 0, "recall": 0}
    gold["metric_vals"] = np.array([float(r[k]) for k in ("dice", "iou", "precision", "recall")], dtype=np.float64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--write-golden", action="store_true")
    args = ap.parse_args()
    if not os.path.isdir(REF):
        raise SystemExit("/root/reference not present: this script only runs in the build container")
    install_shims()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    gold = {}
    check_alp(gold)
    check_fewshot(gold)
    check_dinov2_vs_transformers()
    check_sam_encoder(gold)
    check_sam_decoder(gold)
    check_glue(gold)
    check_amg(gold)
    check_rotate(gold)
    import tempfile
    modules = {}
    with tempfile.TemporaryDirectory() as tmpdir:
        check_orchestration(gold, tmpdir)
        if args.write_golden:
            # the container modules' own forwards and the helper methods by name (tests/test_module_forwards_gpu.py)
            from oracle import make_module_goldens
            make_module_goldens.record(modules, tmpdir)
    if args.write_golden:
        os.makedirs(GOLD, exist_ok=True)
        for fname, arrays in (("reference_outputs.npz", gold), ("reference_modules.npz", modules)):
            path = os.path.join(GOLD, fname)
            if os.path.exists(path):        # leave a file whose arrays are all unchanged alone (no churn in the history)
                old = np.load(path)
                if set(old.files) == set(arrays) and all(np.array_equal(old[k], np.asarray(arrays[k])) for k in arrays):
                    print(f"{path}: unchanged ({len(arrays)} arrays)")
                    continue
            np.savez_compressed(path, **arrays)
            print(f"wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB, {len(arrays)} arrays)")
    print("ALL CHECKS PASSED")


if __name__ == "__main__":
    main()
