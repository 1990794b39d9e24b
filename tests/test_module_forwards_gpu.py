"""GPU: the container modules' own `forward`s and the reference's helper methods by name, against what the REFERENCE's same-named
modules / methods returned for the same seeded weights and inputs (tests/golden/reference_modules.npz, written by
oracle/make_module_goldens.py from /root/reference's vendored segment_anything, models/ProtoSAM.py, models/ProtoMedSAM.py and
util/utils.py in the build container). The hot path never goes through these entry points (it drives the kernels in fused form);
a drop-in user who calls `sam.image_encoder.blocks[i](x)`, `Sam.forward(batched_input, ...)` or `ProtoSAM.get_bbox_per_cc(...)` does.

Tolerances: the image encoder's modules run fp16-operand MFMA GEMMs with fp32 accumulation (2^-11 relative operand rounding): bounded
relative to the output's scale; the decoder side runs at fp32 accuracy; `sigmoid(low_res)` within the north-star 1e-3.
"""
import os
from functools import partial

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_modules.npz")
TOL_PROB = 1e-3


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def _load(mod, dev, seed=None):
    from protosam_amd import synth_cases as gi
    from protosam_amd.synth import synth_state_dict
    mod.load_state_dict(synth_state_dict(mod, gi.MODULE_SEED if seed is None else seed))
    return mod.to(dev).eval()


def _rel(got, ref):
    """max |got - ref| relative to the reference's rms (the natural unit of an fp16-operand product's error)."""
    ref = torch.as_tensor(ref)
    return float((got.detach().float().cpu() - ref).abs().max() / ref.pow(2).mean().sqrt())


@pytest.mark.parametrize("name,ws", [("window", 14), ("global", 0)])
def test_block_forward(dev, gold, name, ws):
    """image_encoder.py:174-193 through `Block.forward` on its own (windowed and global)."""
    from protosam_amd import synth_cases as gi
    from protosam_amd.segment_anything.modeling.image_encoder import Block
    blk = _load(Block(gi.MODULE_DIM, gi.MODULE_HEADS, 4.0, True, partial(torch.nn.LayerNorm, eps=1e-6), torch.nn.GELU, True, True, ws,
                      (64, 64)), dev)
    x = gi.module_block_input().to(dev)
    x0 = x.clone()
    y = blk(x)
    assert y.shape == x.shape and y.dtype == torch.float32 and torch.equal(x, x0)       # (the input is not the residual buffer)
    e = _rel(y[0, 3::8, 5::8], gold[f"block_{name}"])
    print(f"Block.forward ({name}): max err / rms {e:.2e}")
    assert e < 4e-3
    assert torch.equal(blk(x), y)                                                        # cached packing, same result


def test_encoder_attention_forward(dev, gold):
    from protosam_amd import synth_cases as gi
    from protosam_amd.segment_anything.modeling.image_encoder import Attention
    att = _load(Attention(gi.MODULE_DIM, num_heads=gi.MODULE_HEADS, qkv_bias=True, use_rel_pos=True, input_size=(64, 64)), dev)
    y = att(gi.module_block_input().to(dev))
    e = _rel(y[0, 3::8, 5::8], gold["enc_attention"])
    print(f"image_encoder.Attention.forward: max err / rms {e:.2e}")
    assert y.shape == (1, 64, 64, gi.MODULE_DIM) and e < 4e-3
    with pytest.raises(NotImplementedError):
        att(torch.zeros((25, 14, 14, gi.MODULE_DIM), device=dev))


def test_mlp_layernorm2d_patch_embed_forward(dev, gold):
    from protosam_amd import synth_cases as gi
    from protosam_amd.segment_anything.modeling.common import LayerNorm2d, MLPBlock
    from protosam_amd.segment_anything.modeling.image_encoder import PatchEmbed
    mlp = _load(MLPBlock(gi.MODULE_DIM, 4 * gi.MODULE_DIM, torch.nn.GELU), dev)
    e = _rel(mlp(gi.module_mlp_input().to(dev))[::4], gold["mlp_gelu"])
    ln = _load(LayerNorm2d(256), dev)
    y = ln(gi.module_ln2d_input().to(dev))
    e2 = float((y[:, :, ::2, ::2].cpu() - torch.from_numpy(gold["layernorm2d"])).abs().max())
    pe = _load(PatchEmbed((16, 16), (16, 16), in_chans=3, embed_dim=gi.MODULE_DIM), dev)
    z = pe(gi.module_patch_input().to(dev))
    e3 = _rel(z[0, 3::8, 5::8], gold["patch_embed"])
    print(f"MLPBlock (GELU) {e:.2e} of rms, LayerNorm2d {e2:.2e} abs, PatchEmbed {e3:.2e} of rms")
    assert y.shape == (2, 256, 16, 16) and z.shape == (1, 64, 64, gi.MODULE_DIM)
    assert e < 4e-3 and e2 < 1e-5 and e3 < 4e-3


def test_two_way_transformer_forward(dev, gold):
    """transformer.py:62-106,151-182,218-240 through the modules' own forwards (fp32 accuracy: token side fp32, image side x3)."""
    from protosam_amd import synth_cases as gi
    from protosam_amd.segment_anything.modeling.transformer import Attention, TwoWayAttentionBlock, TwoWayTransformer
    emb, ipe, pts = (t.to(dev) for t in gi.module_transformer_inputs())
    tr = _load(TwoWayTransformer(depth=2, embedding_dim=256, num_heads=8, mlp_dim=2048), dev)
    q, k = tr(emb, ipe, pts)
    assert q.shape == (2, 7, 256) and k.shape == (2, 4096, 256)
    eq = float((q.cpu() - torch.from_numpy(gold["twoway_queries"])).abs().max())
    ek = float((k[:, 5::64].cpu() - torch.from_numpy(gold["twoway_keys"])).abs().max())
    print(f"TwoWayTransformer.forward: queries {eq:.2e}, keys {ek:.2e} (abs; LayerNorm-ed outputs of O(1))")
    assert eq < 2e-4 and ek < 2e-4
    blk = _load(TwoWayAttentionBlock(embedding_dim=256, num_heads=8, mlp_dim=2048, skip_first_layer_pe=False), dev)
    keys, kpe = emb.flatten(2).permute(0, 2, 1), ipe.flatten(2).permute(0, 2, 1)
    q, k = blk(queries=pts * 0.5, keys=keys, query_pe=pts, key_pe=kpe)
    eq = float((q.cpu() - torch.from_numpy(gold["twoway_block_queries"])).abs().max())
    ek = float((k[:, 5::64].cpu() - torch.from_numpy(gold["twoway_block_keys"])).abs().max())
    print(f"TwoWayAttentionBlock.forward: queries {eq:.2e}, keys {ek:.2e}")
    assert eq < 2e-4 and ek < 2e-4
    for name, rate, (qq, kk, vv) in (("t2i", 2, (pts, keys + kpe, keys)), ("self", 1, (pts, pts * 0.5, pts)), ("i2t", 2, (keys, pts, pts))):
        a = _load(Attention(256, 8, downsample_rate=rate), dev)
        o = a(q=qq, k=kk, v=vv)
        ref = torch.from_numpy(gold[f"dec_attention_{name}"])
        e = float(((o if o.shape[1] <= 16 else o[:, 5::64]).cpu() - ref).abs().max())
        print(f"transformer.Attention.forward ({name}): {e:.2e} of max |ref| {float(ref.abs().max()):.2f}")
        assert o.shape == qq.shape and e < 1e-4 * max(1.0, float(ref.abs().max()))


def _unpack(bits, shape):
    n = int(np.prod(shape))
    return np.unpackbits(bits)[:n].reshape(shape).astype(bool)


@pytest.mark.parametrize("kind", ["sam_batched", "sam_plain"])
def test_sam_forward(dev, gold, kind):
    """`Sam.forward` / `SamBatched.forward` (sam.py:54-131, :212-290): two images (one zero-padded), points + boxes / points only."""
    from protosam_amd import synth_cases as gi
    from protosam_amd.segment_anything import sam_model_registry
    from protosam_amd.segment_anything.modeling import Sam, SamBatched
    base = sam_model_registry["vit_b"](encoder_depth=gi.ORCH_SAM_DEPTH)
    if kind == "sam_batched":
        sam = SamBatched(base.image_encoder, base.prompt_encoder, base.mask_decoder)
    else:
        sam = Sam(base.image_encoder, base.prompt_encoder, base.mask_decoder)
        sam.postprocess_variant = "nearest"                     # the vendored `Sam` (sam.py:154-160)
    sam = _load(sam, dev, gi.ORCH_SAM_SEED)
    batched = [{k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in rec.items()} for rec in gi.module_sam_forward_input()]
    for mm in (True, False):
        outs = sam(batched, multimask_output=mm)
        assert len(outs) == 2
        for i, o in enumerate(outs):
            pre = f"{kind}_mm{int(mm)}_img{i}"
            shape = tuple(gold[pre + "_shape"])
            assert tuple(o["masks"].shape) == shape and o["masks"].dtype == torch.bool
            perr = (torch.sigmoid(o["low_res_logits"][..., ::2, ::2].cpu()) - torch.sigmoid(torch.from_numpy(gold[pre + "_low"]))).abs().max().item()
            ierr = float((o["iou_predictions"].cpu() - torch.from_numpy(gold[pre + "_iou"])).abs().max())
            ref = _unpack(gold[pre + "_masks"], shape)
            flips = int((o["masks"].cpu().numpy() != ref).sum())
            print(f"{pre}: masks {shape}, max |dprob(low_res)| {perr:.2e}, iou {ierr:.2e}, {flips} of {ref.size} mask pixels differ")
            assert perr < TOL_PROB and ierr < 1e-3 and flips <= max(8, ref.size // 2000)
    if kind == "sam_batched":
        with pytest.raises(KeyError):
            sam([{k: v for k, v in batched[0].items() if k != "image_size"}], multimask_output=True)


def test_protomedsam_segment_all(dev, gold):
    """ProtoMedSAM.segment_all / medsam_inference(query_label) / get_best_mask / get_iou (ProtoMedSAM.py:31-92,224-249)."""
    from protosam_amd import synth_cases as gi
    from protosam_amd.protomedsam import ProtoMedSAM
    m = ProtoMedSAM((1024, 1024), None, f"random:vit_b:{gi.ORCH_SAM_SEED}:{gi.ORCH_SAM_DEPTH}", use_cca=True).to(dev).eval()
    qimg, qlab = gi.module_segment_all_inputs()
    seg, conf = m.segment_all(qimg.to(dev), qlab)
    ref = _unpack(gold["segment_all_mask"], (1024, 1024))
    flips = int((seg.cpu().numpy().astype(bool) != ref).sum())
    cerr = float(np.abs(np.asarray(conf[0]) - gold["segment_all_conf"]).max())
    print(f"segment_all: {flips} px differ of {ref.size}, conf err {cerr:.2e}")
    assert tuple(seg.shape) == (1024, 1024) and flips <= ref.size // 2000 and cerr < 1e-3
    a = np.zeros((8, 8), np.uint8); a[2:6, 2:6] = 1
    b = np.zeros((8, 8), np.uint8); b[3:7, 2:6] = 1
    assert abs(m.get_iou(a, b) - 12 / 20) < 1e-12
    assert m.get_best_mask(np.stack([b * 0, b, a]), torch.from_numpy(a)[None]) is not None
    assert np.array_equal(m.get_best_mask(np.stack([b * 0, b, a]), torch.from_numpy(a)[None]), a)
    assert m.get_best_mask(np.stack([b * 0]), torch.from_numpy(a)[None]) is None
    assert m.get_bbox(np.zeros((4, 4), np.uint8)) is None and list(m.get_bbox(a)) == [2, 2, 5, 5]


def test_connected_components_by_name(dev, gold):
    """util/utils.py:474-541 `get_connected_components` / `cca` on psam_ccl, against the reference's functions (cv2's labelling injected
    from the oracle's restatement when the record was written; label order = raster order of the first pixel in both)."""
    from protosam_amd import synth_cases as gi
    from protosam_amd import utils
    logits = torch.nn.functional.interpolate(gi.orch_coarse_logits(), size=(1024, 1024), mode="bilinear")
    pred = logits.softmax(1).argmax(1)[0].numpy()
    cc, conf = utils.get_connected_components(pred, logits.to(dev), return_conf=True)
    assert cc[0] == int(gold["cc_n"][0]) and np.array_equal(cc[2], gold["cc_stats"])
    assert np.abs(cc[3] - gold["cc_centroids"]).max() < 1e-9
    assert np.array_equal(cc[1][::4, ::4], gold["cc_labels_sub"].astype(np.int32))
    assert np.abs(np.array([float(conf[j]) for j in range(cc[0])]) - gold["cc_conf"]).max() < 1e-5
    cc1 = utils.cca(pred, logits.to(dev), return_cc=True)
    assert cc1[0] == 2 and np.array_equal(cc1[2], gold["cca_stats"]) and np.abs(cc1[3] - gold["cca_centroids"]).max() < 1e-9
    p1, c1 = utils.cca(pred, logits.to(dev), return_conf=True)
    assert int(p1.sum()) == int(gold["cca_pred_sum"][0]) and abs(float(c1) - float(gold["cca_conf"][0])) < 1e-5
    assert utils.get_connected_components(pred, logits.to(dev))[1] is None
    assert abs(utils.get_confidence_from_logits(logits) - utils.get_confidence_from_logits(logits.to(dev))) < 1e-6
    assert utils.need_softmax(logits) and not utils.need_softmax(logits.softmax(1))


def test_protosam_helper_methods(dev, gold):
    """ProtoSAM.get_bbox_per_cc / get_most_conf_points / get_sam_input_points / get_sam_input_mask / predict_w_points_bbox /
    predict_w_masks (ProtoSAM.py:242-289,349-533) by name, against the reference's own methods."""
    from oracle import glue
    from protosam_amd import synth_cases as gi
    from protosam_amd import utils
    from protosam_amd.protosam import ProtoSAM
    spec = f"random:vit_b:{gi.ORCH_SAM_SEED}:{gi.ORCH_SAM_DEPTH}"
    ps = ProtoSAM((1024, 1024), None, spec, use_bbox=True, use_points=True, point_mode="both", use_neg_points=True).to(dev).eval()
    ps.sam.postprocess_variant = "batched"
    logits = torch.nn.functional.interpolate(gi.orch_coarse_logits(), size=(1024, 1024), mode="bilinear")
    output_p = logits.softmax(1)
    pred = output_p.argmax(1)[0].numpy()
    cc, _ = utils.get_connected_components(pred, logits.to(dev), return_conf=True)
    bboxes = ps.get_bbox_per_cc(cc)
    assert np.array_equal(bboxes, gold["ps_bboxes"])
    pts, labs, neg, negl = ps.get_sam_input_points(cc, output_p.to(dev), get_neg_points=True, l=1)
    assert np.array_equal(np.asarray(pts, dtype=np.float64), gold["ps_points"]) and np.array_equal(labs, gold["ps_point_labels"])
    assert np.array_equal(np.stack([np.asarray(n, dtype=np.float64) for n in neg]), gold["ps_neg_points"]) and len(negl) == len(neg)
    loc, cf = ps.get_most_conf_points(output_p[0, 1], torch.tensor(cc[1] == 1).float(), 3)
    assert np.array_equal(loc, gold["ps_top3"]) and np.abs(np.asarray(cf) - gold["ps_top3_conf"]).max() < 1e-7
    m_, l_ = ps.get_sam_input_mask(cc)
    assert np.array_equal(l_, gold["ps_input_mask_labels"]) and np.array_equal(m_.reshape(len(l_), -1).sum(1), gold["ps_input_mask_sums"])
    q1024 = torch.nn.functional.interpolate(gi.orch_query(), size=(1024, 1024), mode="bilinear")
    img = glue.quantise_image(q1024)                                                     # uint8 HWC, ProtoSAM.py:651-660
    masks, scores = ps.predict_w_points_bbox(pts, bboxes, neg, img, pred)
    ref = np.stack([_unpack(b, (1024, 1024)) for b in gold["ps_pwpb_masks"]])
    flips = [int((np.asarray(m) != r).sum()) for m, r in zip(masks, ref)]
    serr = float(np.abs(np.asarray(scores) - gold["ps_pwpb_scores"]).max())
    print(f"predict_w_points_bbox: {len(masks)} components, differing pixels per mask {flips}, scores {serr:.2e}")
    assert len(masks) == len(ref) and max(flips) <= ref[0].size // 2000 and serr < 1e-3
    assert ps.last_stats["low_res"].shape == (len(ref), 256, 256)
    masks, scores = ps.predict_w_masks(m_.copy(), img, 512)
    ref = np.stack([_unpack(b, (1024, 1024)) for b in gold["ps_pwm_masks"]])
    flips = [int((np.asarray(m) != r).sum()) for m, r in zip(masks, ref)]
    serr = float(np.abs(np.asarray(scores) - gold["ps_pwm_scores"]).max())
    print(f"predict_w_masks: differing pixels per mask {flips}, scores {serr:.2e}")
    assert len(masks) == len(ref) and max(flips) <= ref[0].size // 2000 and serr < 1e-3
