"""GPU parity of the glue kernels (connected components, prompts, image hand-off) and of the whole
ProtoSAM.forward path against the CPU oracle on the same seeded support/query pair."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CFG = {"which_model": "dinov2_b14", "cls_name": "grid_proto", "proto_grid_size": 8, "lora": 0, "align": False,
       "debug": False}


def _blobs(seed, H=1024, W=1024, thr=0.15):
    g = torch.Generator().manual_seed(seed)
    f = torch.randn((1, 1, H // 64, W // 64), generator=g)
    f = torch.nn.functional.interpolate(f, size=(H, W), mode="bilinear")[0, 0]
    return f


@pytest.mark.parametrize("seed,thr", [(0, 0.6), (1, 0.2), (2, 1.5), (3, 99.0)])
def test_ccl_table_matches_oracle(dev, seed, thr):
    from oracle import glue
    from protosam_amd import ops
    f = _blobs(seed)
    pred = (f > thr).to(torch.uint8)
    pfg = torch.sigmoid(f)
    ws = ops.CclWorkspace(1024, 1024, 1024, dev)
    fg = pred.sum().to(torch.int32).reshape(1).to(dev)
    ops.ccl(pred.to(dev).contiguous(), pfg.to(dev).contiguous(), ws, fg_sum=fg)
    tab = ws.tab.cpu().numpy()
    labels = ws.labels.cpu().numpy().reshape(1024, 1024)
    n_ref, lab_ref, stats, cent = glue.connected_components_with_stats(pred.numpy())
    assert int(tab[0]) == n_ref - 1 and int(tab[1]) == n_ref - 1
    assert np.array_equal(labels, lab_ref)  # same numbering rule: raster order of the first pixel
    p = pfg.numpy()
    for k in range(n_ref - 1):
        r = tab[ops.CC_HDR + ops.CC_STRIDE * k: ops.CC_HDR + ops.CC_STRIDE * (k + 1)]
        m = lab_ref == k + 1
        assert int(r[0]) == stats[k + 1, 4]
        assert (int(r[3]), int(r[4])) == (stats[k + 1, 0], stats[k + 1, 1])
        assert (int(r[5]), int(r[6])) == (stats[k + 1, 0] + stats[k + 1, 2] - 1, stats[k + 1, 1] + stats[k + 1, 3] - 1)
        assert abs(r[1] / r[0] - cent[k + 1, 0]) < 1e-9 and abs(r[2] / r[0] - cent[k + 1, 1]) < 1e-9
        pt, conf = glue.most_conf_point(p, m)
        assert (int(r[8]), int(r[9])) == (pt[0, 0], pt[0, 1]) and abs(r[10] - conf[0]) < 1e-7
        cref = (p * m).sum() / (pred.sum().item() + 1e-6)
        assert abs(r[7] - cref) < 1e-5 * max(1.0, abs(cref))


def test_image_handoff_quantise(dev):
    """min-max -> uint8 truncation must match numpy bit for bit (ProtoSAM.py:660)."""
    from oracle import glue
    from protosam_amd import ops
    from protosam_amd.synth import synth_pair
    _, _, q, _ = synth_pair(512, seed=5)
    q1024 = torch.nn.functional.interpolate(q, size=(1024, 1024), mode="bilinear")
    ref_u8 = glue.quantise_image(q1024)                     # HWC uint8
    qd = ops.bilinear_nchw(q.to(dev), 1024, 1024)
    torch.testing.assert_close(qd.cpu(), q1024, rtol=1e-6, atol=1e-6)
    mm = ops.minmax(qd, 1)
    u8 = torch.empty((1, 3, 1024, 1024), dtype=torch.uint8, device=dev)
    patches = ops.sam_patchify(qd, mm, 1024, 16, (123.675, 116.28, 103.53), (58.395, 57.12, 57.375), True, u8out=u8)
    got = u8[0].permute(1, 2, 0).cpu().numpy()
    mism = (got != ref_u8).mean()
    assert mism < 1e-4, mism   # identical unless the upstream bilinear differs by an ulp at a truncation boundary
    ref_x = glue.sam_preprocess(got)                        # normalise the SAME uint8 image
    ref_p = ref_x[0].reshape(3, 64, 16, 64, 16).permute(1, 3, 0, 2, 4).reshape(4096, 768)
    torch.testing.assert_close(patches.float().cpu(), ref_p, rtol=2e-3, atol=2e-3)


def _build(dev, sam_spec, dino_depth=None, **kw):
    from protosam_amd.grid_proto_fewshot import FewShotSeg
    from protosam_amd.protosam import ALPNetWrapper, ProtoSAM
    from protosam_amd.synth import synth_state_dict
    cfg = dict(CFG)
    if dino_depth is not None:
        cfg["encoder_depth"] = dino_depth
    alp = FewShotSeg(512, None, cfg)
    alp_sd = synth_state_dict(alp, 1234)
    alp.load_state_dict(alp_sd)
    alp = alp.to(dev).eval()
    model = ProtoSAM(image_size=(1024, 1024), coarse_segmentation_model=ALPNetWrapper(alp), sam_pretrained_path=sam_spec,
                     num_points_for_sam=1, use_sam_trans=True, **kw).to(dev).eval()
    return model, alp_sd


def _dice(a, b):
    a, b = a.float(), b.float()
    tp = (a * b).sum()
    return (2 * tp / (2 * tp + ((1 - a) * b).sum() + (a * (1 - b)).sum() + 1e-8)).item()


@pytest.mark.parametrize("kw", [dict(use_bbox=True, use_points=True, point_mode="both", use_cca=False),
                                dict(use_bbox=True, use_points=True, point_mode="both", use_cca=True),
                                dict(use_bbox=False, use_points=True, point_mode="conf", use_cca=False)])
def test_protosam_forward_vs_oracle(dev, kw):
    from oracle import alp as oalp, dinov2 as odino, glue
    from protosam_amd.protosam import InputFactory, TYPE_ALPNET
    from protosam_amd.synth import synth_pair, synth_state_dict
    sam_depth, dino_depth = 3, 12
    model, alp_sd = _build(dev, f"random:vit_b:1234:{sam_depth}", dino_depth, **kw)
    sam_sd = {k: v.cpu() for k, v in synth_state_dict(model.sam, 1234).items()}
    s_img, s_m, q_img, q_gt = synth_pair(512, seed=0)
    inp = InputFactory.create_input(TYPE_ALPNET, q_img, support_images=[s_img], support_labels=[s_m], isval=True,
                                    val_wsize=2)
    inp.to(dev)
    pred, scores = model(q_img.to(dev), inp, degrees_rotate=0)
    assert pred.shape == (512, 512) and pred.dtype == torch.float32 and pred.is_cuda
    st = model.last_stats
    # --- independent oracle pipeline ---------------------------------------------------------------------------------
    enc_sd = {k[len("encoder."):]: v for k, v in alp_sd.items() if k.startswith("encoder.")}
    enc = lambda im: odino.forward_features(im, enc_sd, "dinov2_b14", depth=dino_depth)["x_norm_patchtokens"]  # noqa
    logits_ref = oalp.fewshot_forward(enc, s_img, s_m, q_img, 512)
    taps = {}
    pred_ref, scores_ref = glue.protosam_forward(q_img, logits_ref, sam_sd, "vit_b", use_bbox=kw["use_bbox"],
                                                 use_points=kw["use_points"], point_mode=kw["point_mode"],
                                                 use_cca=kw["use_cca"], encoder_depth=sam_depth, taps=taps)
    n_ref = 1 if kw["use_cca"] else taps["cc"][0] - 1
    assert st["n_prompts"] == n_ref and len(scores) == len(scores_ref)
    low = st["low_res"][:, st["sel"]].cpu()
    low_ref = torch.stack([l[0] for l in taps["low_res"]])
    perr = (torch.sigmoid(low) - torch.sigmoid(low_ref)).abs().max().item()
    d = _dice(pred.cpu(), pred_ref)
    flips = (pred.cpu() != pred_ref).sum().item()
    print(f"{kw}: comps {n_ref}, max |dprob(low_res)| {perr:.3e}, final Dice {d:.5f}, flipped px {flips}, "
          f"scores {np.abs(np.array(scores) - np.array(scores_ref)).max():.2e}, fg frac {pred_ref.mean():.3f}")
    assert d >= 0.999
    assert perr < 1e-3          # the north-star tolerance on the output probability map
    assert np.abs(np.array(scores, dtype=np.float64) - np.array(scores_ref)).max() < 1e-3


def test_protosam_empty_coarse_mask(dev):
    """Q21: an empty coarse mask returns the 1024x1024 arg-max map and [0] (ProtoSAM.py:612-613)."""
    from protosam_amd.protosam import InputFactory, TYPE_ALPNET
    from protosam_amd.synth import synth_pair
    model, _ = _build(dev, "random:vit_b:1234:1", 1, use_bbox=True, use_points=True, point_mode="both")
    s_img, s_m, q_img, _ = synth_pair(512, seed=0)

    class Zero:  # coarse model that predicts background everywhere
        def __call__(self, inp):
            z = torch.zeros((1, 2, 512, 512), device=dev)
            z[:, 0] = 5.0
            return z
    model.coarse_segmentation_model = Zero()
    inp = InputFactory.create_input(TYPE_ALPNET, q_img, support_images=[s_img], support_labels=[s_m], isval=True,
                                    val_wsize=2)
    pred, scores = model(q_img.to(dev), inp)
    assert pred.shape == (1024, 1024) and pred.dtype == torch.int64 and int(pred.sum()) == 0 and scores == [0]


def test_protosam_constructor_errors():
    from protosam_amd.protosam import ProtoSAM
    with pytest.raises(AssertionError):
        ProtoSAM((1024, 1024), None, "random:vit_b:1:0", use_points=False, use_bbox=False, use_mask=False)
    with pytest.raises(ValueError):
        ProtoSAM((1024, 1024), None, "random:vit_b:1:0", point_mode="nearest")


def test_forward_batch_equals_per_slice(dev):
    """The batched extension gives, slice by slice, what the reference-shaped per-slice forward gives."""
    from protosam_amd.protosam import InputFactory, TYPE_ALPNET
    from protosam_amd.synth import synth_pair
    model, _ = _build(dev, "random:vit_b:1234:2", 2, use_bbox=True, use_points=True, point_mode="both")
    s_img, s_m, q0, _ = synth_pair(512, seed=0)
    _, _, q1, _ = synth_pair(512, seed=3)
    _, _, q2, _ = synth_pair(512, seed=5)
    qs = torch.cat([q0, q1, q2], 0).to(dev)
    inp = InputFactory.create_input(TYPE_ALPNET, qs, support_images=[s_img], support_labels=[s_m], isval=True,
                                    val_wsize=2)
    inp.to(dev)
    batched = model.forward_batch(qs, inp)
    assert len(batched) == 3
    for b in range(3):
        p1, s1 = model(qs[b:b + 1], inp, degrees_rotate=0)
        pb, sb = batched[b]
        assert pb.shape == p1.shape
        assert (pb != p1).sum().item() <= 32, (b, (pb != p1).sum().item())   # (other GEMM kernels per slice than per batch)
        assert len(sb) == len(s1) and np.allclose(np.array(sb, dtype=np.float64), np.array(s1, dtype=np.float64), atol=2e-3)


def test_mixed_support_batch_equals_per_support_batches(dev):
    """forward_batch with a list of (input, n) pairs: slices of DIFFERENT support sets in one batch (a rank of a strong-scaling job,
    or a batch across two z-parts of a scan). Both encoders, connected components, SAM and the decoder do not depend on the support;
    the prototype match is done per support set. With every GEMM on one row-independent kernel (tile 1, separate LayerNorm passes)
    a slice's encoder arithmetic does not depend on what else is in the batch: masks equal to the per-support batches (up to a
    pixel on the threshold: the decoder's token side rounds differently for other prompt counts), scores to fp32 rounding. With the default kernel choice (which follows the batch's row count): within the bound of the batched-vs-per-slice
    test above."""
    from protosam_amd import ops
    from protosam_amd.protosam import InputFactory, TYPE_ALPNET
    from protosam_amd.synth import synth_pair
    model, _ = _build(dev, "random:vit_b:1234:2", 2, use_bbox=True, use_points=True, point_mode="both")
    model.overlap_streams = "0"
    sA, mA, q0, _ = synth_pair(512, seed=0)
    sB, mB, q1, _ = synth_pair(512, seed=3)
    _, _, q2, _ = synth_pair(512, seed=5)
    qs = torch.cat([q0, q2, q1, q2, q0], 0).to(dev)          # slices 0-1: support A, 2-4: support B
    inA = InputFactory.create_input(TYPE_ALPNET, qs[:2], support_images=[sA], support_labels=[mA], isval=True, val_wsize=2)
    inB = InputFactory.create_input(TYPE_ALPNET, qs[2:], support_images=[sB], support_labels=[mB], isval=True, val_wsize=2)
    inA.to(dev); inB.to(dev)
    enc = model.sam.image_encoder
    dino = model.coarse_segmentation_model.model.encoder
    for forced in (True, False):
        fold = (enc.fold_ln, getattr(dino, "fold_ln", None))
        if forced:
            ops.gemm_set_tile(1)
            enc.fold_ln = False
            if fold[1] is not None:
                dino.fold_ln = False
        try:
            mixed = model.forward_batch(qs, [(inA, 2), (inB, 3)])
            sep = model.forward_batch(qs[:2], inA) + model.forward_batch(qs[2:], inB)
        finally:
            ops.gemm_set_tile(0)
            enc.fold_ln = fold[0]
            if fold[1] is not None:
                dino.fold_ln = fold[1]
        assert len(mixed) == len(sep) == 5
        for b, ((pm, sm), (ps, ss)) in enumerate(zip(mixed, sep)):
            assert pm.shape == ps.shape and len(sm) == len(ss)
            if forced:      # (the decoder's token-side linears pick their kernel by the number of prompt sets in the call: scores - and the
                            #  hyper-network vectors - to fp32 rounding, so a mask pixel whose logit is within ~1e-6 of zero may differ)
                assert int((pm != ps).sum()) <= 2, (b, int((pm != ps).sum()))
                assert np.allclose(np.array(sm, dtype=np.float64), np.array(ss, dtype=np.float64), atol=2e-6, rtol=0)
            else:
                assert (pm != ps).sum().item() <= 32
                assert np.allclose(np.array(sm, dtype=np.float64), np.array(ss, dtype=np.float64), atol=2e-3)
    # the supports differ: slice 1 and slice 3 hold the same query image, matched against different prototypes
    assert not torch.equal(mixed[1][0], mixed[3][0]) or mixed[1][1] != mixed[3][1]
    with pytest.raises(NotImplementedError):
        model.forward_batch(qs, [(inA, 2), (inB, 3)], degrees_rotate=10)


def test_forward_batch_overlapped_streams_same_result(dev):
    """`overlap_streams`: the SAM encoder on a second stream next to DINOv2 + ALP + connected components gives the same masks and
    scores as the sequential order; "auto" turns it on only after a call without empty slices."""
    from protosam_amd.protosam import InputFactory, TYPE_ALPNET
    from protosam_amd.synth import synth_pair
    model, _ = _build(dev, "random:vit_b:1234:2", 2, use_bbox=True, use_points=True, point_mode="both")
    s_img, s_m, q0, _ = synth_pair(512, seed=0)
    _, _, q1, _ = synth_pair(512, seed=3)
    qs = torch.cat([q0, q1, q0], 0).to(dev)
    inp = InputFactory.create_input(TYPE_ALPNET, qs, support_images=[s_img], support_labels=[s_m], isval=True, val_wsize=2)
    inp.to(dev)
    outs = {}
    for mode in ("0", "1", "auto", "auto", "auto"):
        model.overlap_streams = mode
        outs.setdefault(mode, []).append(model.forward_batch(qs, inp))
    assert model._dense_run >= 5                               # the last "auto" call ran overlapped (four dense calls before it)
    ref = outs["0"][0]
    for got in (outs["1"][0], outs["auto"][0], outs["auto"][2]):
        for (pa, sa), (pb, sb) in zip(ref, got):
            assert torch.equal(pa, pb) and sa == sb
    # one slice per call (the reference's convention) takes the same two-stream path
    model.overlap_streams = "0"
    p0, s0 = model(qs[1:2], inp)
    model.overlap_streams = "1"
    p1, s1 = model(qs[1:2], inp)
    assert torch.equal(p0, p1) and s0 == s1


@pytest.mark.parametrize("mask_only", [False, True])
def test_forward_batch_skips_sam_for_empty_slices(dev, mask_only):
    """A slice whose coarse mask is empty never reaches SAM (ProtoSAM.py:612-613 returns before set_image): in a batch only
    the non-empty slices are encoded, and every slice still equals its per-slice forward."""
    from protosam_amd.protosam import InputFactory, TYPE_ALPNET
    from protosam_amd.synth import synth_pair
    kw = dict(use_bbox=False, use_points=False, use_mask=True) if mask_only else \
        dict(use_bbox=True, use_points=True, point_mode="both")
    model, _ = _build(dev, "random:vit_b:1234:2", 2, **kw)
    s_img, s_m, q0, _ = synth_pair(512, seed=0)
    _, _, q1, _ = synth_pair(512, seed=3)
    _, _, q2, _ = synth_pair(512, seed=5)
    qs = torch.cat([q0, q1, q2, q0], 0).to(dev)
    inp = InputFactory.create_input(TYPE_ALPNET, qs, support_images=[s_img], support_labels=[s_m], isval=True,
                                    val_wsize=2)
    inp.to(dev)
    real = model.coarse_segmentation_model
    empty = (0, 2)

    class SomeEmpty:   # the real coarse model, but slices 0 and 2 of a 4-batch predict background everywhere
        def __call__(self, cin):
            z = real(cin).clone()
            if z.shape[0] == 4:
                for b in empty:
                    z[b, 0], z[b, 1] = 5.0, -5.0
            return z
    calls = []
    enc = model.sam.image_encoder
    orig = enc.encode_patches
    enc.encode_patches = lambda patches, B, **kw: (calls.append(B), orig(patches, B, **kw))[1]
    try:
        model.coarse_segmentation_model = SomeEmpty()
        batched = model.forward_batch(qs, inp)
        assert calls == [2]                                   # two of four slices went through the image encoder
        for b in empty:
            pb, sb = batched[b]
            assert pb.shape == (1024, 1024) and int(pb.sum()) == 0 and sb == [0]
        model.coarse_segmentation_model = real
        for b in (1, 3):
            p1, s1 = model(qs[b:b + 1], inp)
            pb, sb = batched[b]
            # (the one-slice call takes other GEMM kernels - 128-tile / split-K - than the batch: rounding-level differences
            # flip a few border pixels of the ~20 000-pixel mask)
            assert pb.shape == p1.shape and (pb != p1).sum().item() <= 32
            assert len(sb) == len(s1) and np.allclose(np.array(sb, dtype=np.float64), np.array(s1, dtype=np.float64), atol=2e-3)
        calls.clear()
        model.coarse_segmentation_model = SomeEmpty()
        empty = (0, 1, 2, 3)
        out = model.forward_batch(qs, inp)
        assert calls == [] and all(int(p.sum()) == 0 and sc == [0] for p, sc in out)
    finally:
        enc.encode_patches = orig
        model.coarse_segmentation_model = real


def test_protomedsam_forward_vs_oracle(dev):
    from oracle import alp as oalp, dinov2 as odino, glue
    from protosam_amd.grid_proto_fewshot import FewShotSeg
    from protosam_amd.protomedsam import ProtoMedSAM
    from protosam_amd.protosam import ALPNetWrapper, InputFactory, TYPE_ALPNET
    from protosam_amd.synth import synth_pair, synth_state_dict
    cfg = dict(CFG)
    cfg["encoder_depth"] = 4
    alp = FewShotSeg(512, None, cfg)
    alp_sd = synth_state_dict(alp, 1234)
    alp.load_state_dict(alp_sd)
    alp = alp.to(dev).eval()
    model = ProtoMedSAM((1024, 1024), ALPNetWrapper(alp), "random:vit_b:1234:2", use_cca=True).to(dev).eval()
    sam_sd = {k: v.cpu() for k, v in synth_state_dict(model.medsam, 1234).items()}
    s_img, s_m, q_img, _ = synth_pair(512, seed=0)
    inp = InputFactory.create_input(TYPE_ALPNET, q_img, support_images=[s_img], support_labels=[s_m], isval=True,
                                    val_wsize=2)
    inp.to(dev)
    seg, conf = model(q_img.to(dev), inp)
    assert seg.shape == (512, 512) and seg.dtype == torch.uint8
    enc_sd = {k[len("encoder."):]: v for k, v in alp_sd.items() if k.startswith("encoder.")}
    enc = lambda im: odino.forward_features(im, enc_sd, "dinov2_b14", depth=4)["x_norm_patchtokens"]  # noqa: E731
    logits_ref = oalp.fewshot_forward(enc, s_img, s_m, q_img, 512)
    seg_ref, conf_ref = glue.protomedsam_forward(q_img, logits_ref, sam_sd, "vit_b", use_cca=True, encoder_depth=2)
    d = _dice(seg.cpu(), seg_ref)
    print(f"ProtoMedSAM: Dice {d:.5f}, flipped {(seg.cpu() != seg_ref).sum().item()}, "
          f"conf {float(conf[0].ravel()[0]):.4f} vs {float(conf_ref[0].ravel()[0]):.4f}")
    assert d >= 0.999 and abs(float(conf[0].ravel()[0]) - float(conf_ref[0].ravel()[0])) < 5e-3


@pytest.mark.parametrize("use_cca", [False, True])
def test_protosam_mask_prompts_vs_oracle(dev, use_cca):
    """use_mask=True with points and boxes off (ProtoSAM.py:452-498,664-665): every component's mask becomes a dense
    prompt (nearest 256x256, values 10 / uint8(-8)), the best-scoring of the three masks is kept."""
    from oracle import alp as oalp, dinov2 as odino, glue
    from protosam_amd.protosam import InputFactory, TYPE_ALPNET
    from protosam_amd.synth import synth_pair, synth_state_dict
    sam_depth, dino_depth = 3, 12
    model, alp_sd = _build(dev, f"random:vit_b:1234:{sam_depth}", dino_depth, use_bbox=False, use_points=False,
                           use_mask=True, use_cca=use_cca)
    sam_sd = {k: v.cpu() for k, v in synth_state_dict(model.sam, 1234).items()}
    s_img, s_m, q_img, _ = synth_pair(512, seed=0)
    inp = InputFactory.create_input(TYPE_ALPNET, q_img, support_images=[s_img], support_labels=[s_m], isval=True,
                                    val_wsize=2)
    inp.to(dev)
    pred, scores = model(q_img.to(dev), inp)
    st = model.last_stats
    enc_sd = {k[len("encoder."):]: v for k, v in alp_sd.items() if k.startswith("encoder.")}
    enc = lambda im: odino.forward_features(im, enc_sd, "dinov2_b14", depth=dino_depth)["x_norm_patchtokens"]  # noqa
    logits_ref = oalp.fewshot_forward(enc, s_img, s_m, q_img, 512)
    taps = {}
    pred_ref, scores_ref = glue.protosam_forward(q_img, logits_ref, sam_sd, "vit_b", use_bbox=False, use_points=False,
                                                 use_mask=True, use_cca=use_cca, encoder_depth=sam_depth, taps=taps)
    n_ref = 1 if use_cca else taps["cc"][0] - 1
    assert st["n_prompts"] == n_ref == len(scores) == len(scores_ref)
    low = st["low_res"][:, 1:].cpu()
    low_ref = torch.stack(taps["low_res"])
    perr = (torch.sigmoid(low) - torch.sigmoid(low_ref)).abs().max().item()
    serr = np.abs(np.array(scores, dtype=np.float64) - np.array(scores_ref)).max()
    d = _dice(pred.cpu(), pred_ref)
    print(f"mask prompts (use_cca={use_cca}): comps {n_ref}, max |dprob(low_res)| {perr:.3e}, scores {serr:.2e}, Dice {d:.5f}")
    assert perr < 1e-3 and serr < 1e-3 and d >= 0.999
    # with points or boxes on, the reference overwrites the mask-prompt result (:667-668): use_mask changes nothing
    both, _ = _build(dev, f"random:vit_b:1234:{sam_depth}", dino_depth, use_bbox=True, use_points=True, use_mask=True,
                     point_mode="both", use_cca=use_cca)
    plain, _ = _build(dev, f"random:vit_b:1234:{sam_depth}", dino_depth, use_bbox=True, use_points=True, use_mask=False,
                      point_mode="both", use_cca=use_cca)
    pa, sa = both(q_img.to(dev), inp)
    pb, sb = plain(q_img.to(dev), inp)
    assert torch.equal(pa, pb) and sa == sb


def test_neg_points_kernel_vs_oracle(dev):
    """psam_neg_points: ring (10 x 3x3 dilation minus the component) and global (p_bg >= 0.95) negative points, exact."""
    import scipy.ndimage as ndi
    from oracle import glue
    from protosam_amd import ops
    rng = np.random.RandomState(4)
    H = W = 1024
    for trial, thr in enumerate((0.05, 0.3, 0.6)):
        field = ndi.gaussian_filter(rng.randn(H, W), 18 + 6 * trial)
        field /= np.abs(field).max()
        pred = (field > thr).astype(np.uint8)
        pred[:6, :40] = 1                                    # a component touching the border
        pred[500:503, 500:503] = 1                           # a tiny one
        pfg = np.clip(0.5 + 0.5 * field + 0.02 * rng.randn(H, W), 0.0, 1.0).astype(np.float32)
        prob = torch.from_numpy(np.stack([1.0 - pfg, pfg])[None]).float()
        ws = ops.CclWorkspace(H, W, 256, dev)
        ops.ccl(torch.from_numpy(pred).to(dev), prob[0, 1].to(dev).contiguous(), ws)
        keys = ops.neg_points(ws, prob[0, 0].to(dev).contiguous(), ws.tabs[0], 64).cpu().numpy()
        n = int(ws.tabs[0][1].item())
        lab = ws.labels.view(H, W).cpu().numpy()
        cc = (n + 1, lab, None, None)
        ref = glue.sam_neg_points(cc, prob)
        assert 2 <= n <= 64 and len(ref) == n
        bg = prob[0, 0].numpy()
        glob = ops.decode_point_key(int(keys[0]), W)
        for k in range(n):
            got = [p for p in (ops.decode_point_key(int(keys[1 + k]), W), glob) if p is not None]
            exp = ref[k]
            assert exp is not None and len(got) == len(exp)
            for g, e in zip(got, exp):
                assert (g[0], g[1]) == (int(e[0]), int(e[1])), (trial, k, g, e)
                assert g[2] == bg[g[1], g[0]]
        assert np.all(keys[n + 1:] == 0)
    assert ops.decode_point_key(0, W) is None


def test_protosam_neg_points_vs_oracle(dev):
    """use_neg_points=True (ProtoSAM.py:361-372,395-419,508-511): positive points + ring / global negative points + box."""
    from oracle import alp as oalp, dinov2 as odino, glue
    from protosam_amd.protosam import InputFactory, TYPE_ALPNET
    from protosam_amd.synth import synth_pair, synth_state_dict
    sam_depth, dino_depth = 3, 12
    kw = dict(use_bbox=True, use_points=True, point_mode="both", use_cca=False, use_neg_points=True)
    model, alp_sd = _build(dev, f"random:vit_b:1234:{sam_depth}", dino_depth, **kw)
    sam_sd = {k: v.cpu() for k, v in synth_state_dict(model.sam, 1234).items()}
    s_img, s_m, q_img, _ = synth_pair(512, seed=0)
    inp = InputFactory.create_input(TYPE_ALPNET, q_img, support_images=[s_img], support_labels=[s_m], isval=True,
                                    val_wsize=2)
    inp.to(dev)
    pred, scores = model(q_img.to(dev), inp)
    st = model.last_stats
    enc_sd = {k[len("encoder."):]: v for k, v in alp_sd.items() if k.startswith("encoder.")}
    enc = lambda im: odino.forward_features(im, enc_sd, "dinov2_b14", depth=dino_depth)["x_norm_patchtokens"]  # noqa
    logits_ref = oalp.fewshot_forward(enc, s_img, s_m, q_img, 512)
    taps = {}
    pred_ref, scores_ref = glue.protosam_forward(q_img, logits_ref, sam_sd, "vit_b", use_bbox=True, use_points=True,
                                                 point_mode="both", use_cca=False, use_neg_points=True,
                                                 encoder_depth=sam_depth, taps=taps)
    coords, labels = st["prompts"]
    assert len(coords) == len(taps["neg_points"]) == taps["cc"][0] - 1
    for c, l, neg in zip(coords, labels, taps["neg_points"]):
        got = [tuple(int(v) for v in xy) for xy, lab in zip(c, l) if lab == 0]
        # the fp16 coarse map can move an arg-max by a pixel on a plateau; the labels and counts must agree exactly
        assert len(got) == len(neg) and l.count(0) == len(neg) and l[:2] == [1, 1] and l[-2:] == [2, 3]
        for g, e in zip(got, neg):
            assert abs(g[0] - int(e[0])) <= 2 and abs(g[1] - int(e[1])) <= 2, (got, neg)
    low = st["low_res"][:, st["sel"]].cpu()
    low_ref = torch.stack([l[0] for l in taps["low_res"]])
    perr = (torch.sigmoid(low) - torch.sigmoid(low_ref)).abs().max().item()
    d = _dice(pred.cpu(), pred_ref)
    print(f"neg points: comps {len(coords)}, max |dprob(low_res)| {perr:.3e}, Dice {d:.5f}, "
          f"scores {np.abs(np.array(scores) - np.array(scores_ref)).max():.2e}")
    assert d >= 0.999 and perr < 1e-3
    with pytest.raises(TypeError):
        bad, _ = _build(dev, f"random:vit_b:1234:1", 1, use_bbox=True, use_points=False, use_neg_points=True)
        bad(q_img.to(dev), inp)


def test_config5_medsam_1024_four_classes(dev):
    """BASELINE config 5 in miniature: 1024x1024 inputs (DINOv2 at 1022^2 -> 73x73 grid, 36x36 pooled cells), MedSAM ViT-B
    variant, four classes = four prototype banks of one support image (the reference loops over classes, n_ways == 1,
    grid_proto_fewshot.py:172; validation.py:207). Reduced depths keep the CPU oracle to seconds."""
    from oracle import alp as oalp, dinov2 as odino, glue
    from protosam_amd.grid_proto_fewshot import FewShotSeg
    from protosam_amd.protomedsam import ProtoMedSAM
    from protosam_amd.protosam import ALPNetWrapper, InputFactory, TYPE_ALPNET
    from protosam_amd.synth import ellipse_mask, synth_pair, synth_state_dict
    S = 1024
    cfg = dict(CFG, encoder_depth=2)
    alp = FewShotSeg(S, None, cfg)
    assert alp.config["feature_hw"] == [73, 73] and alp.cls_unit.kernel_size[0] == 9
    alp_sd = synth_state_dict(alp, 1234)
    alp.load_state_dict(alp_sd)
    alp = alp.to(dev).eval()
    model = ProtoMedSAM((1024, 1024), ALPNetWrapper(alp), "random:vit_b:1234:2", use_cca=True).to(dev).eval()
    sam_sd = {k: v.cpu() for k, v in synth_state_dict(model.medsam, 1234).items()}
    s_img, s_m, q_img, _ = synth_pair(S, seed=2)
    enc_sd = {k[len("encoder."):]: v for k, v in alp_sd.items() if k.startswith("encoder.")}
    enc = lambda im: odino.forward_features(im, enc_sd, "dinov2_b14", depth=2)["x_norm_patchtokens"]  # noqa: E731
    # four "organs": the pair's ellipse and three others drawn on the support slice
    masks = [s_m] + [torch.from_numpy(ellipse_mask(S, cy, cx, ry, rx)[None]) for cy, cx, ry, rx in
                     ((0.25, 0.3, 0.1, 0.12), (0.7, 0.72, 0.14, 0.09), (0.3, 0.75, 0.08, 0.15))]
    worst = 1.0
    for ci, m in enumerate(masks):
        inp = InputFactory.create_input(TYPE_ALPNET, q_img, support_images=[s_img], support_labels=[m], isval=True,
                                        val_wsize=2)
        inp.to(dev)
        logits = alp(inp.supp_imgs, inp.fore_mask, inp.back_mask, inp.qry_imgs, True, 2)[0]
        logits_ref = oalp.fewshot_forward(enc, s_img, m, q_img, S)
        perr = (logits.cpu().softmax(1) - logits_ref.softmax(1)).abs().max().item()
        seg, conf = model(q_img.to(dev), inp)
        seg_ref, conf_ref = glue.protomedsam_forward(q_img, logits_ref, sam_sd, "vit_b", use_cca=True, encoder_depth=2)
        assert seg.shape == (S, S) and seg.dtype == torch.uint8
        if seg_ref.sum() == 0:
            assert int(seg.sum()) == 0
            continue
        d = _dice(seg.cpu(), seg_ref)
        worst = min(worst, d)
        print(f"class {ci}: coarse prob err {perr:.2e}, Dice {d:.5f}, fg {int(seg_ref.sum())} px")
        assert perr < 1e-3 and d >= 0.999
    assert worst >= 0.999


@pytest.mark.parametrize("use_cca", [False, True])
def test_coarse_pred_only(dev, use_cca):
    """coarse_pred_only=True (ProtoSAM.py:580-590): the ALPNet argmax map at the query's own size and its mean foreground
    confidence; with use_cca only the most confident component (util/utils.py:496-541) and ITS confidence."""
    from oracle import alp as oalp, dinov2 as odino, glue
    from protosam_amd.protosam import InputFactory, TYPE_ALPNET
    from protosam_amd.synth import synth_pair
    dino_depth = 12
    model, alp_sd = _build(dev, "random:vit_b:1234:1", dino_depth, use_bbox=True, use_points=True, coarse_pred_only=True,
                           use_cca=use_cca)
    s_img, s_m, q_img, _ = synth_pair(512, seed=0)
    inp = InputFactory.create_input(TYPE_ALPNET, q_img, support_images=[s_img], support_labels=[s_m], isval=True,
                                    val_wsize=2)
    inp.to(dev)
    pred, conf = model(q_img.to(dev), inp)
    assert pred.shape == (512, 512) and pred.dtype == torch.int64 and len(conf) == 1
    enc_sd = {k[len("encoder."):]: v for k, v in alp_sd.items() if k.startswith("encoder.")}
    enc = lambda im: odino.forward_features(im, enc_sd, "dinov2_b14", depth=dino_depth)["x_norm_patchtokens"]  # noqa
    logits = oalp.fewshot_forward(enc, s_img, s_m, q_img, 512)
    ref = logits.argmax(1)[0].numpy()
    if not use_cca:
        conf_ref = glue.confidence_from_logits(logits)
    else:
        cc, confs = glue.get_connected_components(ref, logits)
        k = max(confs, key=lambda j: confs[j])
        conf_ref = float(confs[k])
        ref = (cc[1] == k).astype(np.int64) if conf_ref > 0 else ref
    flips = int((pred.cpu().numpy() != ref).sum())
    print(f"coarse_pred_only use_cca={use_cca}: {flips} differing pixels of 262144, conf {conf[0]:.5f} vs {conf_ref:.5f}")
    assert flips <= 40 and abs(float(conf[0]) - conf_ref) < 2e-3


@pytest.mark.parametrize("use_cca", [False, True])
def test_coarse_pred_only_batch_equals_per_slice(dev, use_cca):
    """forward_batch with coarse_pred_only (round 5: one softmax / argmax launch, one connected-components chain and one table copy
    for the batch, nothing of SAM) returns per slice what `forward` returns for it (ProtoSAM.py:580-590)."""
    from protosam_amd.protosam import InputFactory, TYPE_ALPNET
    from protosam_amd.synth import synth_pair
    model, _ = _build(dev, "random:vit_b:1234:1", 2, use_bbox=True, use_points=True, coarse_pred_only=True, use_cca=use_cca)
    s_img, s_m, _, _ = synth_pair(512, seed=0)
    qs = torch.cat([synth_pair(512, seed=sd)[2] for sd in (0, 3, 4)]).to(dev)
    mk = lambda q: InputFactory.create_input(TYPE_ALPNET, q, support_images=[s_img], support_labels=[s_m], isval=True, val_wsize=2)  # noqa: E731
    inp = mk(qs)
    inp.to(dev)
    batched = model.forward_batch(qs, inp)
    assert len(batched) == 3
    for b in range(3):
        inp1 = mk(qs[b:b + 1])
        inp1.to(dev)
        pred, conf = model(qs[b:b + 1], inp1)
        assert batched[b][0].shape == pred.shape and batched[b][0].dtype == pred.dtype
        assert int((batched[b][0] != pred).sum()) <= 8 and abs(float(batched[b][1][0]) - float(conf[0])) < 1e-4


def test_protomedsam_coarse_pred_only(dev):
    from protosam_amd.grid_proto_fewshot import FewShotSeg
    from protosam_amd.protomedsam import ProtoMedSAM
    from protosam_amd.protosam import ALPNetWrapper, InputFactory, TYPE_ALPNET
    from protosam_amd.synth import synth_pair, synth_state_dict
    cfg = dict(CFG, encoder_depth=2)
    alp = FewShotSeg(512, None, cfg)
    alp.load_state_dict(synth_state_dict(alp, 1234))
    alp = alp.to(dev).eval()
    s_img, s_m, q_img, _ = synth_pair(512, seed=0)
    inp = InputFactory.create_input(TYPE_ALPNET, q_img, support_images=[s_img], support_labels=[s_m], isval=True, val_wsize=2)
    inp.to(dev)
    logits = alp(inp.supp_imgs, inp.fore_mask, inp.back_mask, [q_img.to(dev)], True, 2)[0]
    for use_cca in (False, True):
        m = ProtoMedSAM((1024, 1024), ALPNetWrapper(alp), "random:vit_b:1234:1", use_cca=use_cca, coarse_pred_only=True).to(dev).eval()
        pred, conf = m(q_img.to(dev), inp)
        assert pred.shape == (512, 512) and len(conf) == 1 and 0.0 <= float(conf[0]) <= 1.0
        if not use_cca:
            assert torch.equal(pred.bool(), logits.argmax(1)[0].bool())
        else:
            assert int(pred.sum()) <= int(logits.argmax(1)[0].sum())


def _with_env(name, value, fn):
    import os
    old = os.environ.get(name)
    os.environ[name] = value
    try:
        return fn()
    finally:
        if old is None:
            os.environ.pop(name, None)
        else:
            os.environ[name] = old


def test_graph_replay_equals_eager_and_follows_new_weights(dev):
    """The one- / two-slice encoder forwards replay captured HIP graphs (ops.GraphCache): (i) the replay is bit-identical to the eager
    launches (PSAM_HIPGRAPH=0) for DINOv2 `forward_tokens`, SAM `encode_patches` and the whole `ProtoSAM.forward`; (ii) a
    `load_state_dict` after the capture drops the graphs - the next call computes with the NEW weights, equal to an eager run of them
    (a graph holds the addresses of the old weight packs); (iii) a dispatch switch flipped after the capture (`gemm_set_tile`) is part of
    the graph key: the call after it is the switched computation, not a replay of the old kernels."""
    from protosam_amd import ops
    from protosam_amd.protosam import InputFactory, TYPE_ALPNET
    from protosam_amd.synth import synth_pair, synth_state_dict
    model, _ = _build(dev, "random:vit_b:1234:2", 2, use_bbox=True, use_points=True, point_mode="both", use_cca=False)
    s_img, s_m, q_img, _ = synth_pair(512, seed=0)
    q = q_img.to(dev)

    def run():
        inp = InputFactory.create_input(TYPE_ALPNET, q, support_images=[s_img], support_labels=[s_m], isval=True, val_wsize=2)
        inp.to(dev)
        pred, scores = model(q, inp)
        return pred.clone(), model.last_stats["low_res"].clone()
    enc, sam_enc = model.coarse_segmentation_model.model.encoder, model.sam.image_encoder
    for mode in ("auto", "0"):            # graphs are captured on the first call of a shape, replayed on the second
        _with_env("PSAM_HIPGRAPH", mode, run)
    g_pred, g_low = _with_env("PSAM_HIPGRAPH", "auto", run)
    assert enc.__dict__.get("_graphs") is not None and sam_enc.__dict__.get("_graphs") is not None      # (the replay path was taken)
    e_pred, e_low = _with_env("PSAM_HIPGRAPH", "0", run)
    assert torch.equal(g_pred, e_pred) and torch.equal(g_low, e_low)
    tok_g = _with_env("PSAM_HIPGRAPH", "auto", lambda: enc.forward_tokens(q, 504).clone())
    tok_e = _with_env("PSAM_HIPGRAPH", "0", lambda: enc.forward_tokens(q, 504).clone())
    assert torch.equal(tok_g, tok_e)
    # (ii) new weights for both encoders after the capture
    model.sam.load_state_dict(synth_state_dict(model.sam, 4321))
    alp = model.coarse_segmentation_model.model
    alp.load_state_dict(synth_state_dict(alp, 4321))
    assert "_graphs" not in enc.__dict__ and "_graphs" not in sam_enc.__dict__
    n_pred, n_low = _with_env("PSAM_HIPGRAPH", "auto", run)
    n_pred2, n_low2 = _with_env("PSAM_HIPGRAPH", "auto", run)                      # (captured with the new weights, replayed)
    ne_pred, ne_low = _with_env("PSAM_HIPGRAPH", "0", run)
    assert not torch.equal(n_low, g_low)
    assert torch.equal(n_low, ne_low) and torch.equal(n_low2, ne_low) and torch.equal(n_pred, ne_pred)
    # (iii) a dispatch switch after the capture
    ops.gemm_set_tile(1)
    try:
        t1_g = _with_env("PSAM_HIPGRAPH", "auto", lambda: enc.forward_tokens(q, 504).clone())
        t1_e = _with_env("PSAM_HIPGRAPH", "0", lambda: enc.forward_tokens(q, 504).clone())
    finally:
        ops.gemm_set_tile(0)
    assert torch.equal(t1_g, t1_e)
    # (iv) the split-K form of the fp32-residual GEMM shares ONE workspace per device, ordered between streams by a host-tracked event a
    # replayed graph knows nothing of: it is never taken inside a capture. Without the half-tile kernels the one-slice fc2 / proj shapes
    # are split-K candidates; two graph replays on two streams (DINOv2 on one, the SAM encoder on the other) must equal the eager
    # single-stream run of the unsplit kernels.
    ops.gemm_set_option("half_tiles", 0)
    old_ov = model.overlap_streams
    try:
        model.overlap_streams = "1"
        _with_env("PSAM_HIPGRAPH", "auto", run)
        h_pred, h_low = _with_env("PSAM_HIPGRAPH", "auto", run)
        model.overlap_streams = "0"
        ops.gemm_set_option("splitk", 0)
        he_pred, he_low = _with_env("PSAM_HIPGRAPH", "0", run)
    finally:
        model.overlap_streams = old_ov
        ops.gemm_set_option("splitk", 1)
        ops.gemm_set_option("half_tiles", 1)
    assert torch.equal(h_low, he_low) and torch.equal(h_pred, he_pred)
