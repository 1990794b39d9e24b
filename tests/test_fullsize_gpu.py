"""GPU: BASELINE.json configs 3 / 4 / 5 at FULL model depth against the CPU oracle's records
(tests/golden/fullsize_cfg{3,4,5}.npz, written by oracle/make_fullsize_goldens.py in the build container; the oracle itself is
pinned to the reference by oracle/validate_against_reference.py).

  config 3: DINOv2 ViT-B/14 x12 + ALP + SAM ViT-B x12 on the 32-slice MRI-like volume (default flags and use_cca)
  config 4: ... + SAM ViT-H x32 on the 64-slice CT-like volume (the benchmark's workload); also with heavy-tailed SAM weights
  config 5: DINOv2 ViT-B/14 x12 at 1022^2 + MedSAM ViT-B x12, 1024x1024 slice, four classes

The north-star tolerance is asserted as written: |sigmoid(low_res_masks) - reference| <= 1e-3 and the coarse probability map
within 1e-3; the final mask is compared as Dice (thresholds are discontinuities: a handful of border pixels may flip).
Also: the batched volume runner equals the per-slice `ProtoSAM.forward` over the WHOLE volume (slices are independent).
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-3            # the north-star bound: on sigmoid(low_res_masks), and on the predicted-IoU scores `forward` returns beside the mask


def _unpack(bits, S):
    return torch.from_numpy(np.unpackbits(bits)[:S * S].reshape(S, S).astype(np.float32))


def _volume_setup(dev, cfg, wseed=1234, vseed=0):
    from protosam_amd.synth_cases import volume_config
    from protosam_amd.runner import build_protosam, support_set
    from protosam_amd.synth import synth_volume
    sam_type, n, kind, slices, flagsets = volume_config(cfg)
    model, _ = build_protosam(dev, sam_type=sam_type, image_size=512, seed=wseed, heavy_tail=(cfg == 44))
    vol, _ = synth_volume(n, 512, seed=vseed, kind=kind)
    svol, slab = synth_volume(n, 512, seed=vseed + 1, kind=kind)
    sup_imgs, sup_masks = support_set(svol.to(dev), slab.to(dev))
    return model, vol.to(dev), sup_imgs, sup_masks, n, slices, flagsets


@pytest.mark.parametrize("cfg", [3, 4, 44])
def test_config_full_depth_vs_oracle_record(dev, cfg):
    """cfg 44 = config 4 with heavy-tailed SAM ViT-H weights (synth.heavy_tail_sam_: channel scales over 1.5 decades, 'massive
    activation' channels in the hundreds, Student-t output projections): the fp16-operand path is held to the same 1e-3 where
    the residual stream looks like a trained checkpoint's, not only on well-conditioned Gaussian weights."""
    from protosam_amd.metrics import dice
    from protosam_amd.runner import run_slices
    gold = np.load(os.path.join(GOLD, f"fullsize_cfg{cfg}.npz" if cfg != 44 else "fullsize_cfg4_heavytail.npz"))
    model, vol_d, sup_imgs, sup_masks, n, slices, flagsets = _volume_setup(dev, cfg)
    worst = 0.0
    for fname, fl in flagsets.items():
        model.use_cca = fl["use_cca"]
        for z in slices:
            masks, _ = run_slices(model, vol_d, sup_imgs, sup_masks, [z], dev)       # per-slice ProtoSAM.forward
            st = model.last_stats
            k = f"z{z}_{fname}"
            ref_prob = torch.from_numpy(gold[k + "_prob"].astype(np.float32) / 65535.0)
            ref_scores = gold[k + "_scores"]
            assert st["n_prompts"] == ref_prob.shape[0] == len(ref_scores), (st["n_prompts"], ref_prob.shape)
            prob = torch.sigmoid(st["low_res"][:, st["sel"]].cpu())
            perr = (prob - ref_prob).abs().max().item()
            serr = float(np.abs(st["iou"][:, st["sel"]].cpu().numpy() - ref_scores).max())
            ref_mask = _unpack(gold[k + "_mask"], 512)
            d = dice(masks[0].cpu().float(), ref_mask)
            flips = int((masks[0].cpu().float() != ref_mask).sum())
            print(f"config {cfg} {fname} z={z}: {st['n_prompts']} comp, max |dprob(low_res)| {perr:.2e}, scores {serr:.2e}, "
                  f"Dice {d:.5f} ({flips} px)")
            worst = max(worst, perr)
            assert perr <= TOL, (cfg, fname, z, perr)
            assert serr <= TOL and d >= 0.999, (cfg, fname, z, d, flips)            # BASELINE.md's Dice gate
    print(f"config {cfg}: worst max |dprob(low_res)| {worst:.2e} (bound {TOL:.0e})")


@pytest.mark.parametrize("cfg", [4, 44])
def test_batched_step_vs_oracle_record(dev, cfg):
    """The benchmark's own path against the same records: 16 slices of one z-part per `forward_batch` call - the launches then fill
    the CUs and the encoders run with the LayerNorms folded into the assembly GEMM's epilogues (ops.fold_pays), which the
    one-slice calls above do not reach."""
    from protosam_amd.metrics import dice
    from protosam_amd.runner import part_assign, run_slices
    gold = np.load(os.path.join(GOLD, "fullsize_cfg4.npz" if cfg == 4 else "fullsize_cfg4_heavytail.npz"))
    model, vol_d, sup_imgs, sup_masks, n, slices, flagsets = _volume_setup(dev, cfg)
    assert model.sam.image_encoder.fold_ln
    fname = next(k for k, fl in flagsets.items() if not fl["use_cca"])
    model.use_cca = False
    worst = 0.0
    for z in slices:
        part = [y for y in range(n) if part_assign(y, n) == part_assign(z, n)]
        i = part.index(z)
        zs = part[max(0, min(i - 8, len(part) - 16)):][:16]
        masks, _ = run_slices(model, vol_d, sup_imgs, sup_masks, zs, dev, batch=16)
        st = model.last_stats
        b = zs.index(z)
        _, start, cnt = next(sp for sp in st["spans"] if sp[0] == b)
        k = f"z{z}_{fname}"
        ref_prob = torch.from_numpy(gold[k + "_prob"].astype(np.float32) / 65535.0)
        assert cnt == ref_prob.shape[0]
        prob = torch.sigmoid(st["low_res"][start:start + cnt, st["sel"]].cpu())
        perr = (prob - ref_prob).abs().max().item()
        serr = float(np.abs(st["iou"][start:start + cnt, st["sel"]].cpu().numpy() - gold[k + "_scores"]).max())
        d = dice(masks[b].cpu().float(), _unpack(gold[k + "_mask"], 512))
        print(f"config {cfg} batched z={z} ({len(zs)} slices per call): max |dprob(low_res)| {perr:.2e}, scores {serr:.2e}, Dice {d:.5f}")
        worst = max(worst, perr)
        assert perr <= TOL and serr <= TOL and d >= 0.999
    print(f"config {cfg} batched: worst max |dprob(low_res)| {worst:.2e} (bound {TOL:.0e})")


@pytest.mark.parametrize("cfg", [3, 4])
def test_volume_runner_equals_per_slice_forward(dev, cfg):
    """`run_slices` with 16-slice batches over the whole 32- / 64-slice volume == one `ProtoSAM.forward` per slice."""
    from protosam_amd.metrics import dice
    from protosam_amd.runner import run_slices
    model, vol_d, sup_imgs, sup_masks, n, _, _ = _volume_setup(dev, cfg)
    zs = list(range(n))
    batched, st_b = run_slices(model, vol_d, sup_imgs, sup_masks, zs, dev, batch=16)
    batched = batched.clone()
    single, st_s = run_slices(model, vol_d, sup_imgs, sup_masks, zs, dev, batch=1)
    assert st_b == st_s                                                   # same number of prompt sets on every slice
    diff = (batched != single).flatten(1).sum(1).cpu()
    ds = [dice(batched[z].cpu().float(), single[z].cpu().float()) for z in zs]
    print(f"config {cfg}: {n} slices, worst per-slice difference {int(diff.max())} px, worst Dice {min(ds):.5f}, "
          f"{sum(st_b)} prompt sets")
    # (the batched run picks other GEMM kernels than the per-slice one - since round 3 the half-tile assembly kernels, which add the
    # bias before the products instead of after them: equally valid fp32 rounding, amplified through 12 / 32 blocks into logit
    # differences of a few 1e-4, which flip border pixels where |logit| is that small. Each path is held to 1e-3 against the oracle
    # record above; here: at most ~0.1 % of a slice's pixels and Dice >= 0.995 between the two paths - with the folded LayerNorm the
    # worst slice of config 4 has 80 border pixels of a 9 200-pixel mask different, Dice 0.9957)
    assert int(diff.max()) <= 256 and all(int(diff[z]) <= 32 or ds[z] >= 0.995 for z in zs)


def _prompt_difference(prompts, ref, tie_bits):
    """None: the HIP path's prompts of a slice equal the oracle's record ([components, 8] = most confident point, centroid, box).
    "tolerance": they differ only in ways a coarse probability map within 1e-3 of the oracle's allows - a most confident point on a pixel
    the ORACLE's probabilities put within 1e-3 of the component's maximum (`tie_bits`: packed 1024 x 1024 mask), a centroid / box edge
    one pixel off. Anything else: a description of the difference (a failure)."""
    if prompts is None:
        return None if len(ref) == 0 else "no prompts"
    mine = np.array([np.asarray(c, dtype=np.float64).reshape(-1)[:8] for c in prompts[0]], dtype=np.float64).reshape(-1, 8)
    if mine.shape != ref.shape:
        return f"{mine.shape[0]} prompt sets, the oracle has {ref.shape[0]}"
    # equal: the same most confident pixel, the same box, the centroid within 0.05 px (a coarse-mask pixel or two at p = 0.5 flip in
    # most slices - the coarse probabilities agree to 1e-4, not to the bit - and move the mean of 70 000 pixel coordinates by ~0.005 px)
    if np.abs(mine[:, [0, 1, 4, 5, 6, 7]] - ref[:, [0, 1, 4, 5, 6, 7]]).max() <= 1e-3 and np.abs(mine[:, 2:4] - ref[:, 2:4]).max() <= 0.05:
        return None
    tie = np.unpackbits(tie_bits)[:1024 * 1024].reshape(1024, 1024)
    for k in range(len(ref)):
        if np.abs(mine[k, :2] - ref[k, :2]).max() > 1e-3 and not tie[int(mine[k, 1]), int(mine[k, 0])]:
            return f"component {k}: most confident point {mine[k, :2]} is not among the oracle's near-ties (its own: {ref[k, :2]})"
        if np.abs(mine[k, 2:] - ref[k, 2:]).max() > 1.0:
            return f"component {k}: centroid / box {mine[k, 2:]} vs {ref[k, 2:]}"
    return "tolerance"


def _redecode_with_oracle_prompts(model, vol_d, z, ref_prompts):
    """The HIP image encoder + prompt encoder + mask decoder of slice z fed the ORACLE's recorded prompts (`z<z>_prompts`: most confident
    point, centroid, box per component) through `ProtoSAM.predict_w_points_bbox` - for a slice whose own most-confident point is another
    one of the oracle's near-ties (an arg-max, not arithmetic), this is the like-for-like comparison of everything downstream of the
    prompt choice. -> (sigmoid(low_res) [n,256,256], scores [n], final mask [S,S] bool)."""
    from protosam_amd import ops
    S = vol_d.shape[-1]
    q = vol_d[z][None, None].expand(1, 3, S, S).contiguous().float()
    qd = ops.bilinear_nchw(q, 1024, 1024)                                    # ProtoSAM.py:592-593, then :651-660
    mm = ops.minmax(qd, 1)
    u8 = torch.empty((1, 3, 1024, 1024), dtype=torch.uint8, device=qd.device)
    sam = model.sam
    ops.sam_patchify(qd, mm, 1024, 16, sam._mean_host, sam._std_host, True, u8out=u8)
    img = u8[0].permute(1, 2, 0).cpu().numpy()
    pts = ref_prompts[:, :4].astype(np.float64).reshape(-1, 2, 2)
    boxes = ref_prompts[:, 4:8].astype(np.float64)
    masks, scores = model.predict_w_points_bbox(pts, boxes, [None] * len(pts), img, None)
    prob = torch.sigmoid(torch.from_numpy(np.asarray(model.last_stats["low_res"])))
    union = np.logical_or.reduce(np.stack(masks))
    return prob, np.asarray(scores), torch.from_numpy(union[::1024 // S, ::1024 // S].copy())      # 'nearest' to S x S (:674)


# (config, weight seed, volume seed): the round-4 records (every slice, weights 1234, volume 0) and, round 5, two more weight draws and
# one more volume per configuration (config 3: every slice; config 4: every 4th) - oracle/make_fullsize_goldens.py --wseed / --vseed
# round 6: a fourth weight draw / third volume of config 4 (weights 99, volume 7), and config 4 with HEAVY-TAILED SAM-H weights (44:
# synth.heavy_tail_sam_ - channel scales over 1.5 decades, massive-activation channels, Student-t projections) over every 4th slice
VOLUME_VARIANTS = [(3, 1234, 0), (4, 1234, 0), (3, 777, 0), (3, 4242, 0), (3, 1234, 5), (4, 777, 0), (4, 4242, 0), (4, 1234, 5),
                   (4, 99, 7), (44, 1234, 0)]


@pytest.mark.parametrize("cfg,wseed,vseed", VOLUME_VARIANTS)
def test_whole_volume_vs_oracle_masks(dev, cfg, wseed, vseed):
    """The slices of config 3 / config 4 volumes against the oracle's records (tests/golden/fullvolume_cfg{3,4}[_w<seed>_v<seed>].npz)
    on SEVERAL weight draws and volumes (the 1e-3 has to be a property of the implementation, not of one draw), for BOTH HIP paths -
    one ProtoSAM.forward per slice, and 16-slice forward_batch calls (LayerNorm folded into the GEMMs, batches that span z-parts):
      * sigmoid(low_res_masks) of the kept token (every 4th pixel) and the scores within the north-star 1e-3, the same number of
        prompt sets, on both paths;
      * the final masks: mean Dice over the slices >= 0.999 (BASELINE.md section 4's gate with the caller's aggregation:
        validation_protosam.py computes the metric of :169-185 per slice and averages, :399-403);
      * per slice, every flipped pixel has to be one the tolerance explains: at most as many as the oracle's record counts pixels whose
        up-sampled logit lies within 4e-3 of the threshold (`z<z>_amb`: a probability error of 1e-3 is a logit error of 4e-3 there, and
        only such a pixel can change sign). Records without the count (none since round 5) fall back to Dice >= 0.998 or <= 32 px;
      * the DISCRETE decisions in between - which pixel is a component's most confident one (an arg-max over near-ties: softmax
        saturates inside a confident region), where a component's box ends - are compared with the oracle's (`z<z>_prompts`). Equal
        prompts: the bounds above apply in full. A different most-confident point has to be one the oracle's own probabilities put
        within 1e-3 of the component's maximum (`z<z>_tie`), a box edge / centroid may move by one pixel (a border pixel of the coarse
        mask at p = 0.5); such a slice's own result answers ANOTHER prompt than the record's (Dice >= 0.985 is asked of it, and at most
        one slice in four may be of that kind - measured 0 ... 5 of 32; the reference's own CPU and CUDA runs differ the same way,
        torch.topk leaves the order of equal values unspecified), so the slice is decoded AGAIN from the oracle's recorded prompts
        (`_redecode_with_oracle_prompts`: the HIP encoder and decoder through `ProtoSAM.predict_w_points_bbox`) and THAT result is
        held to the same bounds as every other slice: no slice of the eight records is exempt from the 1e-3."""
    from oracle.make_fullsize_goldens import volume_record_name
    from protosam_amd.metrics import dice
    from protosam_amd.runner import run_slices
    gold = np.load(os.path.join(GOLD, volume_record_name(cfg, wseed, vseed)))
    model, vol_d, sup_imgs, sup_masks, n, _, _ = _volume_setup(dev, cfg, wseed, vseed)
    model.use_cca = False
    zs = [int(z) for z in gold["zs"]] if "zs" in gold.files else list(range(n))
    redecoded = {}                               # z -> the re-decoding from the oracle's prompts (the same for both paths)
    paths = (("per-slice", 1), ("batched", 16))
    if (cfg, wseed, vseed) == (4, 1234, 0):      # bench.py's step since round 6: 32 slices per forward_batch call (a call spans two z-parts)
        paths += (("batched-32", 32),)
    for name, batch in paths:
        dices, worst_p, worst_s, flips, bad, amb_used, moved = [], 0.0, 0.0, 0, [], 0.0, []
        step = batch                             # (one call per slice on the per-slice path: its last_stats hold that slice's logits)
        for i in range(0, len(zs), step):
            chunk = zs[i:i + step]
            masks, st = run_slices(model, vol_d, sup_imgs, sup_masks, chunk, dev, batch=batch)
            masks = masks.cpu()
            per = model.last_stats
            low = iou = sel = None               # (a call whose slices are all empty has no logits: nothing of an earlier call is reused)
            if "low_res" in per:
                low, iou, sel = per["low_res"].cpu(), per["iou"].cpu(), per["sel"]
            for b, z in enumerate(chunk):
                ref = _unpack(gold[f"z{z}_mask"], 512)
                d = dice(masks[b].float(), ref)
                f = int((masks[b].float() != ref).sum())
                if f"z{z}_prompts" in gold.files:
                    stb = per["per_slice"][b] if batch > 1 else per
                    why = _prompt_difference(stb.get("prompts"), gold[f"z{z}_prompts"], gold[f"z{z}_tie"])
                    if why is not None:
                        assert why == "tolerance", (name, z, why)
                        assert d >= 0.985, (name, z, d)
                        moved.append(z)
                        # ... and everything downstream of the prompt choice, from the ORACLE's prompts, under the full bounds
                        if z not in redecoded:
                            redecoded[z] = _redecode_with_oracle_prompts(model, vol_d, z, gold[f"z{z}_prompts"])
                        p_re, s_re, m_re = redecoded[z]
                        ref_scores = gold[f"z{z}_scores"]
                        assert p_re.shape[0] == len(ref_scores) == st[b], (name, z, p_re.shape, len(ref_scores), st[b])
                        d, f = dice(m_re.float(), ref), int((m_re.float() != ref).sum())
                        dices.append(d)
                        flips = max(flips, f)
                        amb = int(gold[f"z{z}_amb"][0])
                        amb_used = max(amb_used, f / max(amb, 1))
                        if f > amb:
                            bad.append((z, d, f, amb, "re-decoded"))
                        refp = torch.from_numpy(gold[f"z{z}_prob4"].astype(np.float32) / 65535.0)
                        worst_p = max(worst_p, (p_re[..., ::4, ::4] - refp).abs().max().item())
                        worst_s = max(worst_s, float(np.abs(s_re - ref_scores).max()))
                        continue
                dices.append(d)
                flips = max(flips, f)
                if f"z{z}_amb" in gold.files:
                    amb = int(gold[f"z{z}_amb"][0])
                    amb_used = max(amb_used, f / max(amb, 1))
                    if f > amb:
                        bad.append((z, d, f, amb))
                elif d < 0.998 and f > 32:
                    bad.append((z, d, f))
                ref_scores = gold[f"z{z}_scores"]
                assert st[b] == len(ref_scores), (name, z, st[b], len(ref_scores))
                if f"z{z}_prob4" in gold.files:
                    _, start, cnt = next(sp for sp in per["spans"] if sp[0] == b) if batch > 1 else (0, 0, low.shape[0])
                    refp = torch.from_numpy(gold[f"z{z}_prob4"].astype(np.float32) / 65535.0)
                    p = torch.sigmoid(low[start:start + cnt, sel])[..., ::4, ::4]
                    worst_p = max(worst_p, (p - refp).abs().max().item())
                    worst_s = max(worst_s, float(np.abs(iou[start:start + cnt, sel].numpy() - ref_scores).max()))
        below = sum(1 for d in dices if d < 0.999)
        print(f"config {cfg} weights {wseed} volume {vseed} {name}: {len(zs)} slices, mean Dice {np.mean(dices):.5f}, worst {min(dices):.5f}, "
              f"most flipped pixels {flips} (at most {amb_used:.2f} of a slice's tolerance-explained count), {below} slice(s) below 0.999, "
              f"max |dprob(low_res)| {worst_p:.2e}, scores {worst_s:.2e}")
        if moved:
            print(f"    slices whose prompts moved within the tolerance band (another most-confident point among near-ties / a box edge by one pixel), "
                  f"re-decoded from the oracle's prompts and included above: {moved}")
        assert len(moved) <= max(1, len(zs) // 4), (name, moved)
        assert worst_p <= TOL and worst_s <= TOL, (name, worst_p, worst_s)
        assert np.mean(dices) >= 0.999 and not bad, (name, np.mean(dices), bad)


def test_config4_slice_vs_reference_full_depth_record(dev):
    """The HIP path against the REFERENCE's own full-depth run (no oracle in between): tests/golden/reference_fullsize_vith.npz holds
    what the reference's 32-block `sam_model_registry["vit_h"]` + `SamPredictor.set_image / predict` returned for slice 32 of config
    4 (oracle/make_reference_fullsize.py, which also asserts the oracle equal to it: embedding and logits bit for bit)."""
    from protosam_amd.runner import run_slices
    gold = np.load(os.path.join(GOLD, "reference_fullsize_vith.npz"))
    model, vol_d, sup_imgs, sup_masks, n, _, _ = _volume_setup(dev, 4)
    model.use_cca = False
    z = int(gold["z"][0])
    masks, _ = run_slices(model, vol_d, sup_imgs, sup_masks, [z], dev)
    st = model.last_stats
    low_ref = torch.from_numpy(gold["low_res"])                      # [n, 3, 256, 256] logits of the three mask tokens
    assert st["n_prompts"] == low_ref.shape[0]
    # (the decoder keeps four mask tokens; multimask_output=True hands back tokens 1..3, mask_decoder.py:107-112)
    perr = (torch.sigmoid(st["low_res"][:, 1:4].cpu()) - torch.sigmoid(low_ref)).abs().max().item()
    serr = float(np.abs(st["iou"][:, 1:4].cpu().numpy() - gold["iou"]).max())
    print(f"config 4 z={z} vs the reference's ViT-H x32 run: max |dprob(low_res)| over the three tokens {perr:.2e}, iou {serr:.2e}")
    assert perr <= TOL and serr <= TOL
    d = (masks[0].cpu().float() != _unpack(gold["mask"], 512)).sum().item()
    assert d <= 64, d


def test_config5_full_depth_vs_oracle_record(dev):
    from protosam_amd.synth_cases import cfg5_inputs
    from protosam_amd.grid_proto_fewshot import FewShotSeg
    from protosam_amd.metrics import dice
    from protosam_amd.protomedsam import ProtoMedSAM
    from protosam_amd.protosam import ALPNetWrapper, InputFactory, TYPE_ALPNET
    from protosam_amd.runner import ALP_CFG
    from protosam_amd.synth import synth_state_dict
    gold = np.load(os.path.join(GOLD, "fullsize_cfg5.npz"))
    S = 1024
    alp = FewShotSeg(S, None, dict(ALP_CFG))
    alp.load_state_dict(synth_state_dict(alp, 1234))
    alp = alp.to(dev).eval()
    assert alp.config["feature_hw"] == [73, 73]
    model = ProtoMedSAM((1024, 1024), ALPNetWrapper(alp), "random:vit_b:1234", use_cca=True).to(dev).eval()
    s_img, s_masks, q_img = cfg5_inputs()
    ran = 0
    for ci, m in enumerate(s_masks):
        inp = InputFactory.create_input(TYPE_ALPNET, q_img, support_images=[s_img], support_labels=[m], isval=True,
                                        val_wsize=2)
        inp.to(dev)
        logits = alp(inp.supp_imgs, inp.fore_mask, inp.back_mask, inp.qry_imgs, True, 2)[0]
        cp = logits.float().softmax(1)[0, 1, ::4, ::4].cpu()
        cerr = (cp - torch.from_numpy(gold[f"class{ci}_coarse_p"].astype(np.float32) / 65535.0)).abs().max().item()
        seg, conf = model(q_img.to(dev), inp)
        ref_mask = _unpack(gold[f"class{ci}_mask"], S)
        assert seg.shape == (S, S) and seg.dtype == torch.uint8
        if f"class{ci}_prob" not in gold.files:                           # empty coarse mask for this class
            assert int(seg.sum()) == 0 and cerr <= TOL
            continue
        ran += 1
        prob = torch.sigmoid(model.last_stats["low_res"][:, 0].cpu())
        perr = (prob - torch.from_numpy(gold[f"class{ci}_prob"].astype(np.float32) / 65535.0)).abs().max().item()
        d = dice(seg.cpu().float(), ref_mask)
        ce = float(np.abs(np.asarray(conf[0]) - gold[f"class{ci}_conf"]).max())
        flips = int((seg.cpu().float() != ref_mask).sum())
        print(f"config 5 class {ci}: coarse prob err {cerr:.2e}, max |dprob(low_res)| {perr:.2e}, conf err {ce:.2e}, "
              f"Dice {d:.5f} ({flips} px of {int(ref_mask.sum())})")
        # (the smallest organ is ~900 px: a handful of threshold flips on its border is already 0.002 of Dice)
        assert cerr <= TOL and perr <= TOL and ce <= TOL and (d >= 0.998 or flips <= 8)
    assert ran == 4


def test_config5_forward_classes_equals_per_class_forward(dev):
    """`ProtoMedSAM.forward_classes` (one DINOv2 forward of the query shared by the four prototype banks, one MedSAM encoder
    forward, one batched decoder call) against the four separate `forward()` calls the reference's multi-class loop makes
    (validation.py:207), and against the oracle record."""
    from protosam_amd.synth_cases import cfg5_inputs
    from protosam_amd.grid_proto_fewshot import FewShotSeg
    from protosam_amd.metrics import dice
    from protosam_amd.protomedsam import ProtoMedSAM
    from protosam_amd.protosam import ALPNetWrapper, InputFactory, TYPE_ALPNET
    from protosam_amd.runner import ALP_CFG
    from protosam_amd.synth import synth_state_dict
    gold = np.load(os.path.join(GOLD, "fullsize_cfg5.npz"))
    S = 1024
    alp = FewShotSeg(S, None, dict(ALP_CFG))
    alp.load_state_dict(synth_state_dict(alp, 1234))
    alp = alp.to(dev).eval()
    model = ProtoMedSAM((1024, 1024), ALPNetWrapper(alp), "random:vit_b:1234", use_cca=True).to(dev).eval()
    s_img, s_masks, q_img = cfg5_inputs()
    s_d, q_d, m_d = s_img.to(dev), q_img.to(dev), [m.to(dev) for m in s_masks]
    res = model.forward_classes(q_d, s_d, m_d)
    low_all = model.last_stats["low_res"].cpu()
    assert len(res) == 4
    k = 0
    for ci, m in enumerate(s_masks):
        inp = InputFactory.create_input(TYPE_ALPNET, q_img, support_images=[s_img], support_labels=[m], isval=True, val_wsize=2)
        inp.to(dev)
        seg1, conf1 = model(q_d, inp)
        seg, conf = res[ci]
        assert seg.shape == seg1.shape
        flips = int((seg.cpu() != seg1.cpu()).sum())
        ref_mask = _unpack(gold[f"class{ci}_mask"], S)
        d = dice(seg.cpu().float(), ref_mask)
        if f"class{ci}_prob" in gold.files:
            perr = (torch.sigmoid(low_all[k, 0]) - torch.from_numpy(gold[f"class{ci}_prob"].astype(np.float32) / 65535.0)).abs().max().item()
            cerr = float(np.abs(np.asarray(conf[0]) - np.asarray(conf1[0])).max())
            k += 1
            print(f"class {ci}: {flips} px differ from the per-class forward, conf diff {cerr:.2e}; vs oracle record: max |dprob| {perr:.2e}, Dice {d:.5f}")
            assert flips <= 4 and cerr <= 1e-4 and perr <= TOL
        else:
            assert int(seg.sum()) == 0
    # the second call (same support objects) reuses the cached banks and gives the same result
    res2 = model.forward_classes(q_d, s_d, m_d)
    assert all(torch.equal(a[0], b[0]) for a, b in zip(res, res2))
