"""GPU parity of the automatic mask generator and `SamWrapper` against the CPU oracle
(oracle/amg.py, pinned to the vendored reference by oracle/validate_against_reference.py) and the reference's recorded
records in tests/golden/reference_outputs.npz."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_outputs.npz")


@pytest.mark.parametrize("variant,H,W", [(0, 1024, 1024), (1, 1024, 1024), (1, 768, 1024), (2, 1024, 640)])
def test_mask_stats_and_binarize_kernels(dev, variant, H, W):
    """Counts / boxes / binary masks computed from the 256x256 logits equal the same reductions of the materialised
    up-sampling (`psam_mask_upsample`, itself checked against F.interpolate in test_sam_gpu.py) EXACTLY."""
    from protosam_amd import ops
    g = torch.Generator().manual_seed(variant * 7 + H)
    low = torch.randn((6, 4, 256, 256), generator=g) * 2.0
    low = F.avg_pool2d(low, 9, 1, 4) * 6.0                      # smooth blobs with values on both sides of +-1
    low[1, 2] = -5.0                                            # an empty mask
    low[2, 1] = 5.0                                             # a full one
    low[3, 3, :, :] = -5.0
    low[3, 3, 100:103, 40:41] = 3.0                             # a tiny island
    low = low.to(dev).contiguous()
    thr, off = 0.0, 1.0
    stats = ops.mask_stats(low, 1, 3, 1024, H, W, variant, thr, off).cpu().numpy()
    up = ops.mask_upsample(low, 1024, variant)[:, 1:, :H, :W].reshape(-1, H, W)
    assert stats.shape == (18, 8)
    hi = (up > thr + off).sum((1, 2)).cpu().numpy()
    lo = (up > thr - off).sum((1, 2)).cpu().numpy()
    m = up > thr
    area = m.sum((1, 2)).cpu().numpy()
    np.testing.assert_array_equal(stats[:, 0], hi)
    np.testing.assert_array_equal(stats[:, 1], lo)
    np.testing.assert_array_equal(stats[:, 2], area)
    for p in range(18):
        if area[p] == 0:
            assert stats[p, 3:7].tolist() == [2**31 - 1, 2**31 - 1, -1, -1]
            continue
        ys, xs = torch.nonzero(m[p], as_tuple=True)
        assert stats[p, 3:7].tolist() == [int(xs.min()), int(ys.min()), int(xs.max()), int(ys.max())]
    assert area[1 * 3 + 1] == 0 and area[2 * 3 + 0] == H * W
    # binarise a subset (indices into low.view(-1, 256, 256)) and score it against a label
    idx = torch.tensor([1, 6, 9, 15, 23], dtype=torch.int32, device=dev)
    label = (torch.rand((H, W), generator=g) > 0.6).to(torch.uint8).to(dev)
    out, counts = ops.mask_binarize(low, idx, 1024, H, W, variant, thr, label=label)
    full = ops.mask_upsample(low, 1024, variant).reshape(-1, 1024, 1024)[idx.long(), :H, :W] > thr
    assert out.dtype == torch.uint8 and torch.equal(out.bool(), full)
    lb = label.bool()[None]
    exp = torch.stack([(full & lb).sum((1, 2)), (full & ~lb).sum((1, 2)), (~full & lb).sum((1, 2))], 1)
    assert torch.equal(counts, exp)
    out2, none = ops.mask_binarize(low, idx, 1024, H, W, variant, thr)
    assert none is None and torch.equal(out2, out)


@pytest.fixture(scope="module")
def amg_setup(dev):
    """SAM vit_b (2 encoder blocks, synthetic weights) + the oracle's view of every candidate of the 8x8 grid."""
    from oracle import amg as oamg
    from protosam_amd import synth_cases as gi
    from protosam_amd.sam_wrapper import SamWrapper
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    gold = np.load(GOLD)
    t_iou, t_stab = (float(v) for v in gold["amg_thresholds"])
    args = dict(gi.AMG_ARGS, pred_iou_thresh=t_iou, stability_score_thresh=t_stab)
    w = SamWrapper({"model_type": "vit_b", "sam_checkpoint": f"random:{gi.AMG_SEED}:{gi.AMG_ENCODER_DEPTH}",
                    "generator_args": args}).to(dev)
    sd = {k: v.detach().cpu() for k, v in w.sam.state_dict().items()}
    img, label = gi.amg_case()
    taps = {}
    okw = dict(encoder_depth=gi.AMG_ENCODER_DEPTH, **args)
    best, bi, ious, anns = oamg.sam_wrapper_forward(img, label, sd, taps=taps, **okw)
    return dict(w=w, sd=sd, img=img, label=label, gold=gold, taps=taps, best=best, bi=bi, ious=ious, anns=anns, args=args)


def test_candidates_vs_oracle(dev, amg_setup):
    """Every candidate (no filtering, no suppression): predicted IoU, stability score, area and box vs the oracle."""
    from protosam_amd.segment_anything import SamAutomaticMaskGenerator
    s = amg_setup
    g = SamAutomaticMaskGenerator(s["w"].sam, **dict(s["args"], pred_iou_thresh=0.0, stability_score_thresh=0.0))
    cand = g._candidates(s["img"])
    plane = cand["plane"].cpu().numpy()
    k = (plane // 4) * 3 + plane % 4 - 1                                         # position in the oracle's flatten(0, 1)
    assert sorted(k.tolist()) == list(range(192))
    iou_o, stab_o = s["taps"]["iou_all"].numpy()[k], s["taps"]["stab_all"].numpy()[k]
    e_iou = np.abs(cand["iou_preds"] - iou_o).max()
    e_stab = np.abs(cand["stability_score"] - stab_o).max()
    print(f"192 candidates: predicted IoU max err {e_iou:.2e}, stability max err {e_stab:.2e}")
    assert e_iou < 5e-3 and e_stab < 5e-3
    assert np.all(np.diff(cand["iou_preds"]) <= 0)                              # NMS order = decreasing score
    assert (cand["boxes"] == np.array([0, 0, 1023, 1023])).all()                 # noise-like masks span the image


def test_generate_vs_reference_records(dev, amg_setup):
    """`generate` with the golden thresholds keeps the same candidates as the REFERENCE run recorded in the fixture."""
    from protosam_amd.segment_anything import SamAutomaticMaskGenerator
    s, gold = amg_setup, amg_setup["gold"]
    g = SamAutomaticMaskGenerator(s["w"].sam, **s["args"])
    anns = g.generate(s["img"])
    assert len(anns) == len(gold["amg_pred_iou"]) == len(s["anns"])
    got = {tuple(a["point_coords"][0]) for a in anns}
    ref = {tuple(p) for p in gold["amg_points"].tolist()}
    assert got == ref
    # records agree one by one once both lists are ordered by (point, predicted IoU)
    o = sorted(range(len(anns)), key=lambda i: (anns[i]["point_coords"][0], -anns[i]["predicted_iou"]))
    r = sorted(range(len(anns)), key=lambda i: (gold["amg_points"][i].tolist(), -gold["amg_pred_iou"][i]))
    for i, j in zip(o, r):
        a = anns[i]
        assert set(a) == {"segmentation", "area", "bbox", "predicted_iou", "point_coords", "stability_score", "crop_box"}
        assert abs(a["predicted_iou"] - gold["amg_pred_iou"][j]) < 5e-3
        assert abs(a["stability_score"] - gold["amg_stability"][j]) < 5e-3
        assert a["bbox"] == gold["amg_bbox"][j].tolist() and a["crop_box"] == [0, 0, 1024, 1024]
        assert abs(a["area"] - int(gold["amg_area"][j])) <= 0.004 * 1024 * 1024
        assert a["segmentation"].dtype == bool and a["segmentation"].shape == (1024, 1024)
        assert int(a["segmentation"].sum()) == a["area"]
        assert isinstance(a["area"], int) and isinstance(a["predicted_iou"], float)
    # default suppression: all boxes coincide, so only the top-scoring record survives (as in the reference run)
    g7 = SamAutomaticMaskGenerator(s["w"].sam, **dict(s["args"], box_nms_thresh=0.7))
    a7 = g7.generate(s["img"])
    assert len(a7) == 1 and a7[0]["point_coords"] == [gold["amg_points"][0].tolist()]
    # uncompressed RLE output decodes to the same mask
    from protosam_amd.segment_anything.utils.amg import area_from_rle, rle_to_mask
    gr = SamAutomaticMaskGenerator(s["w"].sam, **dict(s["args"], box_nms_thresh=0.7, output_mode="uncompressed_rle"))
    ar = gr.generate(s["img"])
    assert np.array_equal(rle_to_mask(ar[0]["segmentation"]), a7[0]["segmentation"])
    assert area_from_rle(ar[0]["segmentation"]) == a7[0]["area"]


def test_sam_wrapper_forward(dev, amg_setup):
    """SamWrapper.forward: the proposal with the best IoU against the label, vs the oracle and the reference fixture."""
    from protosam_amd.protosam import InputFactory, SamWrapperWrapper, TYPE_SAM
    s, gold = amg_setup, amg_setup["gold"]
    out = s["w"](s["img"], s["label"])
    assert out.dtype == bool and out.shape == (1024, 1024)
    st = s["w"].last_stats
    ious_o = np.array([float(v) for v in s["ious"]])
    srt = np.sort(ious_o)[::-1]
    print(f"best IoU oracle {srt[0]:.5f} (runner-up {srt[1]:.5f}), GPU {st['best_iou']:.5f}, {st['n_masks']} proposals")
    assert abs(st["best_iou"] - srt[0]) < 2e-3
    ref_best = np.unpackbits(gold["amg_best_mask_bits"]).reshape(1024, 1024).astype(bool)
    if srt[0] - srt[1] > 4e-3:                                                   # an unambiguous winner
        dice = 2.0 * (out & ref_best).sum() / (out.sum() + ref_best.sum())
        flips = int((out != ref_best).sum())
        print(f"best mask vs reference: Dice {dice:.5f}, {flips} differing pixels")
        assert dice > 0.995
        assert np.array_equal(ref_best, s["best"]) or int((ref_best != s["best"]).sum()) <= 4
    # device-side IoU counts == numpy get_iou on the downloaded masks
    from protosam_amd.sam_wrapper import get_iou
    cand, masks, counts = s["w"].mask_generator.generate_device(s["img"], torch.from_numpy(s["label"]).to(dev))
    m = masks.cpu().numpy()
    for i in range(len(m)):
        assert abs(get_iou(m[i], s["label"]) - st["ious"][i]) < 1e-12
    # the ModelWrapper view used by ProtoSAM: 2-class "logits" on the device
    inp = InputFactory.create_input(TYPE_SAM, torch.zeros((1, 3, 8, 8)).add_(torch.arange(8.0)), gts=torch.zeros((8, 8)))
    assert inp.image.shape == (8, 8, 3) and inp.image.dtype == np.uint8 and inp.image_labels.shape == (8, 8)
    inp.image, inp.image_labels = s["img"], s["label"]
    lg = SamWrapperWrapper(s["w"])(inp)
    assert lg.is_cuda and lg.shape == (1, 2, 1024, 1024)
    assert torch.equal(lg[0, 1].bool().cpu(), torch.from_numpy(out)) and torch.equal(lg[0, 0], 1 - lg[0, 1])
    # error behaviour
    with pytest.raises(ValueError):
        s["w"](s["img"], np.zeros((512, 512), np.uint8))
    with pytest.raises(ValueError):
        s["w"](s["img"], np.full((1024, 1024), 3, np.uint8))


@pytest.mark.parametrize("M,N,K,G", [(7, 256, 256, 1), (45, 128, 256, 1), (1792, 256, 256, 1), (1792, 2048, 256, 1),
                                     (1792, 256, 2048, 1), (100, 4, 256, 1), (256, 32, 256, 4), (33, 70, 128, 1),
                                     (18, 256, 256, 1), (9, 2048, 256, 1), (17, 256, 128, 1), (5, 256, 2048, 1), (31, 130, 64, 1), (8, 32, 256, 4)])
def test_small_linear_both_paths(dev, M, N, K, G):
    """fp32 token-side linear: the one-wave-per-column kernel (M < 32) and the fp32-MFMA tile kernel (M >= 32) against a
    float64 matmul, with the fused x + x2, ReLU, residual and grouped (hyper-network) forms."""
    from protosam_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn((G, M, K), generator=g)
    x2 = torch.randn((G, M, K), generator=g)
    W = torch.randn((G, N, K), generator=g) / K ** 0.5
    b = torch.randn((G, N), generator=g)
    r = torch.randn((G, M, N), generator=g)
    xd, x2d, Wd, bd, rd = (t.to(dev).contiguous() for t in (x, x2, W, b, r))
    for act, use_x2, use_r in ((0, False, False), (1, True, True), (0, True, False), (1, False, True)):
        ref = (x.double() + (x2.double() if use_x2 else 0)) @ W.double().transpose(1, 2) + b.double()[:, None]
        if act:
            ref = ref.clamp_min(0)
        if use_r:
            ref = ref + r.double()
        out = torch.full((G, M, N), float("nan"), device=dev)
        ops.small_linear(xd, Wd, bd, out=out, act=act, resid=rd if use_r else None, x2=x2d if use_x2 else None, G=G, M=M,
                         N=N, K=K, xg=M * K, wg=N * K, bg=N, yg=M * N, ldx=K, ldy=N)
        err = (out.cpu().double() - ref).abs().max().item()
        assert err < 2e-5 * (K / 256) ** 0.5 + 1e-5, (act, use_x2, use_r, err)


# ---- crop layers, arbitrary image sizes, small-region removal (automatic_mask_generator.py:194-380) ------------------------
def test_plane_stats_kernel(dev):
    """`psam_plane_stats`: the three counts, the box and the binary mask of full-resolution logit planes, exactly."""
    from protosam_amd import ops
    g = torch.Generator().manual_seed(11)
    for n, H, W in ((5, 30, 32), (3, 48, 44), (2, 257, 1023), (1, 1, 1)):
        x = torch.randn((n, H, W), generator=g) * 2
        if H > 8:
            x = F.avg_pool2d(x[None], 5, 1, 2)[0] * 4
            x[0] = -5.0
            x[0, 3:5, 7] = 2.0
        xd = x.to(dev).contiguous()
        st, m = ops.plane_stats(xd, 0.25, 1.0)
        st = st.cpu().numpy()
        ref = x > 0.25
        assert torch.equal(m.cpu().bool(), ref)
        np.testing.assert_array_equal(st[:, 0], (x > 1.25).sum((1, 2)).numpy())
        np.testing.assert_array_equal(st[:, 1], (x > -0.75).sum((1, 2)).numpy())
        np.testing.assert_array_equal(st[:, 2], ref.sum((1, 2)).numpy())
        for p in range(n):
            if st[p, 2] == 0:
                continue
            ys, xs = torch.nonzero(ref[p], as_tuple=True)
            assert st[p, 3:7].tolist() == [int(xs.min()), int(ys.min()), int(xs.max()), int(ys.max())]
        st2, none = ops.plane_stats(xd, 0.25, 1.0, binarize=False)
        assert none is None and torch.equal(st2.cpu(), torch.from_numpy(st))


def test_remove_small_regions_on_device(dev):
    """Hole / island removal through `psam_ccl` == the oracle's restatement of utils/amg.py:267-291, bit for bit."""
    from oracle import amg as oamg
    from protosam_amd import ops
    from protosam_amd.segment_anything import SamAutomaticMaskGenerator
    g = SamAutomaticMaskGenerator.__new__(SamAutomaticMaskGenerator)
    rng = np.random.default_rng(3)
    for H, W, p in ((48, 44, 0.55), (40, 64, 0.3), (33, 17, 0.8), (16, 16, 0.0), (16, 16, 1.0)):
        mask = rng.random((H, W)) < p
        if 0 < p < 1:
            mask[10:30, 5:15] = True
            mask[12, 7] = False                                               # a one-pixel hole
        # the second workspace has fewer table rows than the mask has regions: the overflow path must give the same masks
        for cw in (ops.CclWorkspace(H, W, 4096, dev), ops.CclWorkspace(H, W, 4, dev)):
            for mode in ("holes", "islands"):
                for thr in (1, 6, 50, 10 ** 6):
                    ref, ch_ref = oamg.remove_small_regions(mask, thr, mode)
                    got, ch = g._remove_small_regions(torch.from_numpy(mask.astype(np.uint8)).to(dev), thr, mode, cw)
                    assert ch == ch_ref, (H, W, mode, thr, cw.cap)
                    assert np.array_equal(got.cpu().numpy().astype(bool), ref), (H, W, mode, thr, cw.cap)


@pytest.fixture(scope="module")
def amg_crop_setup(dev):
    from protosam_amd import synth_cases as gi
    from protosam_amd.sam_wrapper import SamWrapper
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    w = SamWrapper({"model_type": "vit_b", "sam_checkpoint": f"random:{gi.AMG_SEED}:{gi.AMG_ENCODER_DEPTH}",
                    "generator_args": dict(gi.AMG_ARGS)}).to(dev)
    sd = {k: v.detach().cpu() for k, v in w.sam.state_dict().items()}
    return dict(w=w, sd=sd, img=gi.amg_small_case(), gold=np.load(GOLD))


def _record_key(pt, crop, piou):
    return (tuple(int(v) for v in crop), tuple(round(float(v), 6) for v in pt), -float(piou))


def test_generate_crops_and_small_regions_vs_reference(dev, amg_crop_setup):
    """crop_n_layers = 1 + min_mask_region_area on the 48 x 44 image, no suppression: one record per candidate of every crop
    (64 * 3 from the image, 4 * 16 * 3 from the layer-1 crops), against the REFERENCE's recorded run (amgc_all_*)."""
    from protosam_amd import synth_cases as gi
    from protosam_amd.segment_anything import SamAutomaticMaskGenerator
    s, gold = amg_crop_setup, amg_crop_setup["gold"]
    H, W = s["img"].shape[:2]
    g = SamAutomaticMaskGenerator(s["w"].sam, **dict(gi.AMG_CROP_ARGS, crop_nms_thresh=1.0))
    anns = g.generate(s["img"])
    n = len(gold["amgc_all_pred_iou"])
    assert len(anns) == n == 384
    ref_masks = np.unpackbits(gold["amgc_all_mask_bits"], axis=1)[:, :H * W].reshape(n, H, W).astype(bool)
    o = sorted(range(n), key=lambda i: _record_key(anns[i]["point_coords"][0],
                                                   [anns[i]["crop_box"][0], anns[i]["crop_box"][1]], 0))
    r = sorted(range(n), key=lambda i: _record_key(gold["amgc_all_points"][i], gold["amgc_all_crop_box"][i][:2], 0))
    # three records per (crop, point): pair them by predicted IoU inside each group
    worst_px, worst_iou, worst_stab, exact = 0, 0.0, 0.0, 0
    for lo in range(0, n, 3):
        a3 = sorted(o[lo:lo + 3], key=lambda i: -anns[i]["predicted_iou"])
        r3 = sorted(r[lo:lo + 3], key=lambda i: -gold["amgc_all_pred_iou"][i])
        for i, j in zip(a3, r3):
            a = anns[i]
            assert a["point_coords"][0] == pytest.approx(gold["amgc_all_points"][j].tolist(), abs=1e-9)
            assert a["crop_box"] == gold["amgc_all_crop_box"][j].tolist()
            worst_iou = max(worst_iou, abs(a["predicted_iou"] - float(gold["amgc_all_pred_iou"][j])))
            worst_stab = max(worst_stab, abs(a["stability_score"] - float(gold["amgc_all_stability"][j])))
            d = int((a["segmentation"] != ref_masks[j]).sum())
            worst_px = max(worst_px, d)
            exact += d == 0
            assert a["segmentation"].shape == (H, W) and int(a["segmentation"].sum()) == a["area"]
            if d == 0:
                assert a["bbox"] == gold["amgc_all_bbox"][j].tolist() and a["area"] == int(gold["amgc_all_area"][j])
    n_crop = sum(1 for a in anns if a["crop_box"] != [0, 0, W, H])
    print(f"{n} records ({n_crop} from layer-1 crops): {exact} masks identical to the reference's, worst {worst_px} px, "
          f"predicted IoU err {worst_iou:.2e}, stability err {worst_stab:.2e}")
    assert n_crop == 192 and worst_iou < 5e-3
    # fp16 encoder vs the fp32 reference: a logit near the threshold may flip a pixel, and a flipped pixel may move a region
    # across the area threshold (6 px) - bounded, and most masks are identical
    assert exact >= 0.8 * n and worst_px <= 24


def test_generate_crops_default_suppression(dev, amg_crop_setup):
    """Default crop_nms_thresh: cross-crop NMS prefers the masks of smaller crops (scores = 1 / crop area, :208-218)."""
    from protosam_amd import synth_cases as gi
    from protosam_amd.segment_anything import SamAutomaticMaskGenerator
    s, gold = amg_crop_setup, amg_crop_setup["gold"]
    g = SamAutomaticMaskGenerator(s["w"].sam, **gi.AMG_CROP_ARGS)
    anns = g.generate(s["img"])
    ref_crops = sorted(map(tuple, gold["amgc_nms_crop_box"].tolist()))
    print(f"{len(anns)} records after cross-crop NMS (reference {len(ref_crops)}): crops {[a['crop_box'] for a in anns]}")
    assert sorted(tuple(a["crop_box"]) for a in anns) == ref_crops


def test_generate_any_image_size_vs_oracle(dev, amg_crop_setup):
    """crop_n_layers = 0 on an image that is NOT at the model's input size (set_image resizes it, masks come back at the
    image's size) against the oracle run live."""
    from oracle import amg as oamg
    from protosam_amd import synth_cases as gi
    from protosam_amd.segment_anything import SamAutomaticMaskGenerator
    s = amg_crop_setup
    kw = dict(points_per_side=4, points_per_batch=16, box_nms_thresh=1.0, pred_iou_thresh=0.0, stability_score_thresh=0.0)
    anns = SamAutomaticMaskGenerator(s["w"].sam, **kw).generate(s["img"])
    # (min_mask_region_area = 1 removes nothing - no region is smaller than one pixel - and selects the oracle's general path)
    ref = oamg.generate(s["img"], s["sd"], encoder_depth=gi.AMG_ENCODER_DEPTH, crop_n_layers=0, min_mask_region_area=1, **kw)
    assert len(anns) == len(ref) == 48
    key = lambda a: (a["point_coords"][0], -a["predicted_iou"])  # noqa: E731
    worst = 0
    for a, b in zip(sorted(anns, key=key), sorted(ref, key=key)):
        assert a["point_coords"] == b["point_coords"] and a["crop_box"] == b["crop_box"]
        assert abs(a["predicted_iou"] - b["predicted_iou"]) < 5e-3
        worst = max(worst, int((a["segmentation"] != b["segmentation"]).sum()))
    print(f"48 records on a 48 x 44 image: worst {worst} differing pixels per mask")
    assert worst <= 24
