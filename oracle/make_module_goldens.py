"""Records what the REFERENCE's container modules and helper methods return when they are called on their own
(tests/golden/reference_modules.npz), for the GPU tests of the same-named modules / methods of protosam_amd
(tests/test_module_forwards_gpu.py). Test infrastructure only; runs ONLY in the build container, where /root/reference exists:

  python oracle/make_module_goldens.py            (or: python oracle/validate_against_reference.py --write-golden, which calls record())

Reference code executed (the vendored segment_anything under /root/reference/models, behind the shims of
oracle/validate_against_reference.py):
  modeling/image_encoder.py  Block.forward :174-193 (a windowed and a global block), Attention.forward :235-251, PatchEmbed :402-406
  modeling/common.py         MLPBlock.forward :25-26, LayerNorm2d.forward :38-43
  modeling/transformer.py    TwoWayTransformer.forward :62-106, TwoWayAttentionBlock.forward :151-182, Attention.forward :218-240
  modeling/sam.py            Sam.forward :54-131, SamBatched.forward :212-290
  models/ProtoMedSAM.py      segment_all :224-249, medsam_inference :31-66, get_best_mask :79-92, get_bbox_per_cc :109-120
  models/ProtoSAM.py         get_bbox_per_cc :242-264, get_most_conf_points :266-289, get_sam_input_points :349-450,
                             get_sam_input_mask :452-466, predict_w_masks :468-498, predict_w_points_bbox :500-533
  util/utils.py              get_connected_components :474-494, cca :496-541 (cv2 absent: the restated labelling is injected)
Weights: protosam_amd.synth.synth_state_dict (seeded by parameter name); inputs: protosam_amd.synth_cases. Large outputs are stored
subsampled (the slices are written next to the arrays' names below and repeated in the test).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden", "reference_modules.npz")


def _pack(mask):
    return np.packbits(np.asarray(mask).astype(bool))


def record(gold, tmpdir):
    """Fills `gold` (name -> numpy array). The shims of validate_against_reference.install_shims() must be installed."""
    from functools import partial
    import oracle.validate_against_reference as var
    var._install_cv2_restatements()
    from protosam_amd import synth_cases as gi
    from protosam_amd.synth import synth_state_dict
    from segment_anything.modeling.common import LayerNorm2d, MLPBlock
    from segment_anything.modeling.image_encoder import Attention, Block, PatchEmbed
    from segment_anything.modeling.transformer import Attention as TAttention, TwoWayAttentionBlock, TwoWayTransformer
    seed = gi.MODULE_SEED
    D, H = gi.MODULE_DIM, gi.MODULE_HEADS
    f32 = lambda t: t.detach().numpy().astype(np.float32)  # noqa: E731
    print("stand-alone module forwards of the reference -> tests/golden/reference_modules.npz")
    with torch.no_grad():
        x = gi.module_block_input()
        for name, ws in (("window", 14), ("global", 0)):
            blk = Block(dim=D, num_heads=H, mlp_ratio=4.0, qkv_bias=True, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6),
                        act_layer=torch.nn.GELU, use_rel_pos=True, window_size=ws, input_size=(64, 64)).eval()
            blk.load_state_dict(synth_state_dict(blk, seed))
            gold[f"block_{name}"] = f32(blk(x)[0, 3::8, 5::8])                 # [8,8,768]
        att = Attention(D, num_heads=H, qkv_bias=True, use_rel_pos=True, input_size=(64, 64)).eval()
        att.load_state_dict(synth_state_dict(att, seed))
        gold["enc_attention"] = f32(att(x)[0, 3::8, 5::8])
        mlp = MLPBlock(D, 4 * D, torch.nn.GELU).eval()
        mlp.load_state_dict(synth_state_dict(mlp, seed))
        gold["mlp_gelu"] = f32(mlp(gi.module_mlp_input())[::4])                # [75,768]
        ln = LayerNorm2d(256).eval()
        ln.load_state_dict(synth_state_dict(ln, seed))
        gold["layernorm2d"] = f32(ln(gi.module_ln2d_input())[:, :, ::2, ::2])  # [2,256,8,8]
        pe = PatchEmbed(kernel_size=(16, 16), stride=(16, 16), in_chans=3, embed_dim=D).eval()
        pe.load_state_dict(synth_state_dict(pe, seed))
        gold["patch_embed"] = f32(pe(gi.module_patch_input())[0, 3::8, 5::8])
        # two-way transformer and its parts
        emb, ipe, pts = gi.module_transformer_inputs()
        tr = TwoWayTransformer(depth=2, embedding_dim=256, num_heads=8, mlp_dim=2048).eval()
        tr.load_state_dict(synth_state_dict(tr, seed))
        q, k = tr(emb, ipe, pts)
        gold["twoway_queries"], gold["twoway_keys"] = f32(q), f32(k[:, 5::64])  # [2,7,256], [2,64,256]
        blk = TwoWayAttentionBlock(embedding_dim=256, num_heads=8, mlp_dim=2048, skip_first_layer_pe=False).eval()
        blk.load_state_dict(synth_state_dict(blk, seed))
        keys = emb.flatten(2).permute(0, 2, 1)
        kpe = ipe.flatten(2).permute(0, 2, 1)
        q, k = blk(queries=pts * 0.5, keys=keys, query_pe=pts, key_pe=kpe)
        gold["twoway_block_queries"], gold["twoway_block_keys"] = f32(q), f32(k[:, 5::64])
        for name, rate, (qq, kk, vv) in (("t2i", 2, (pts, keys + kpe, keys)), ("self", 1, (pts, pts * 0.5, pts)), ("i2t", 2, (keys, pts, pts))):
            a = TAttention(256, 8, downsample_rate=rate).eval()
            a.load_state_dict(synth_state_dict(a, seed))
            o = a(q=qq, k=kk, v=vv)
            gold[f"dec_attention_{name}"] = f32(o if o.shape[1] <= 16 else o[:, 5::64])

        # Sam.forward (vendored `Sam`: nearest post-processing, sam.py:154-160) and SamBatched.forward (what the vendored registry builds)
        var._truncate_vendored_registry(gi.ORCH_SAM_DEPTH)
        from segment_anything import sam_model_registry
        from segment_anything.modeling import Sam
        samb = sam_model_registry["vit_b"]().eval()
        sam_sd = synth_state_dict(samb, gi.ORCH_SAM_SEED)
        samb.load_state_dict(sam_sd)
        sam_plain = Sam(samb.image_encoder, samb.prompt_encoder, samb.mask_decoder).eval()
        for name, model in (("sam_batched", samb), ("sam_plain", sam_plain)):
            for mm in (True, False):
                outs = model(gi.module_sam_forward_input(), multimask_output=mm)
                for i, o in enumerate(outs):
                    gold[f"{name}_mm{int(mm)}_img{i}_low"] = f32(o["low_res_logits"][..., ::2, ::2])
                    gold[f"{name}_mm{int(mm)}_img{i}_iou"] = f32(o["iou_predictions"])
                    gold[f"{name}_mm{int(mm)}_img{i}_masks"] = _pack(o["masks"].numpy())
                    gold[f"{name}_mm{int(mm)}_img{i}_shape"] = np.array(o["masks"].shape)

        # ProtoMedSAM.segment_all (three masks of the whole-image box, best IoU against the label)
        import models.ProtoMedSAM as ref_pm
        import models.ProtoSAM as ref_ps
        ckpt = os.path.join(tmpdir, "sam_vit_b_synth_modules.pth")
        torch.save(sam_sd, ckpt)
        med = ref_pm.ProtoMedSAM((1024, 1024), None, ckpt, use_cca=True).eval()
        qimg, qlab = gi.module_segment_all_inputs()
        seg, conf = med.segment_all(qimg, qlab)
        gold["segment_all_mask"], gold["segment_all_conf"] = _pack(seg.numpy()), np.asarray(conf[0], dtype=np.float32)
        print(f"  segment_all: fg {int(seg.sum())} of label {int(qlab.sum())}, conf {np.asarray(conf[0]).ravel()}")

        # ProtoSAM's helper methods on the orchestration case's coarse logits (use_neg_points: the ring + global points too)
        import util.utils as ref_utils
        logits = torch.nn.functional.interpolate(gi.orch_coarse_logits(), size=(1024, 1024), mode="bilinear")
        output_p = logits.softmax(1)
        pred = np.array(output_p.argmax(1)[0])
        cc, conf = ref_utils.get_connected_components(pred, logits, return_conf=True)
        gold["cc_n"] = np.array([cc[0]])
        gold["cc_stats"], gold["cc_centroids"] = np.asarray(cc[2]), np.asarray(cc[3])
        gold["cc_labels_sub"] = np.asarray(cc[1])[::4, ::4].astype(np.int16)
        gold["cc_conf"] = np.array([float(conf[j]) for j in range(cc[0])])
        cc1 = ref_utils.cca(pred, logits, return_cc=True)
        gold["cca_stats"], gold["cca_centroids"] = np.asarray(cc1[2]), np.asarray(cc1[3])
        p1, c1 = ref_utils.cca(pred, logits, return_conf=True)
        gold["cca_pred_sum"], gold["cca_conf"] = np.array([int(p1.sum())]), np.array([float(c1)])
        orig_cpu = torch.Tensor.cpu                       # (device semantics of `.cpu()`: a copy - see validate_against_reference.py)
        torch.Tensor.cpu = lambda self, *a, **k: orig_cpu(self, *a, **k).clone()
        try:
            ps = ref_ps.ProtoSAM((1024, 1024), None, ckpt, use_bbox=True, use_points=True, point_mode="both", use_neg_points=True).eval()
            gold["ps_bboxes"] = np.asarray(ps.get_bbox_per_cc(cc))
            pts_, labs_, neg_, negl_ = ps.get_sam_input_points(cc, output_p, get_neg_points=True, l=1)
            gold["ps_points"], gold["ps_point_labels"] = np.asarray(pts_, dtype=np.float64), np.asarray(labs_)
            gold["ps_neg_points"] = np.stack([np.asarray(n, dtype=np.float64) for n in neg_])
            loc, cf = ps.get_most_conf_points(output_p[0, 1], torch.tensor(cc[1] == 1).float(), 3)
            gold["ps_top3"], gold["ps_top3_conf"] = np.asarray(loc), np.asarray(cf)
            m_, l_ = ps.get_sam_input_mask(cc)
            gold["ps_input_mask_sums"], gold["ps_input_mask_labels"] = m_.reshape(len(l_), -1).sum(1), np.asarray(l_)
            from oracle import glue
            q1024 = torch.nn.functional.interpolate(gi.orch_query(), size=(1024, 1024), mode="bilinear")
            img = glue.quantise_image(q1024)
            masks, scores = ps.predict_w_points_bbox(pts_, gold["ps_bboxes"], neg_, img, pred, return_logits=False)
            gold["ps_pwpb_masks"] = np.stack([_pack(m) for m in masks])
            gold["ps_pwpb_scores"] = np.asarray(scores, dtype=np.float32)
            masks, scores = ps.predict_w_masks(m_.copy(), img, 512)
            gold["ps_pwm_masks"] = np.stack([_pack(m) for m in masks])
            gold["ps_pwm_scores"] = np.asarray(scores, dtype=np.float32)
        finally:
            torch.Tensor.cpu = orig_cpu
    return gold


def main():
    import tempfile
    import oracle.validate_against_reference as var
    if not os.path.isdir(var.REF):
        raise SystemExit("/root/reference not present: this script only runs in the build container")
    var.install_shims()
    import matplotlib
    matplotlib.use("Agg")
    torch.manual_seed(0)
    torch.set_num_threads(8)
    gold = {}
    with tempfile.TemporaryDirectory() as tmpdir:
        record(gold, tmpdir)
    np.savez_compressed(GOLD, **gold)
    print(f"wrote {GOLD} ({os.path.getsize(GOLD) / 1e6:.2f} MB, {len(gold)} arrays)")


if __name__ == "__main__":
    main()
