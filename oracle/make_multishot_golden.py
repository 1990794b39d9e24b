"""Reference records for the two non-default settings of the path: n_shots > 1 (FewShotSeg.forward, grid_proto_fewshot.py:244-266) and,
below, `num_points_for_sam` > 1 (ProtoSAM.get_most_conf_points, ProtoSAM.py:266-289, 376-387). Runs ONLY in the build container:
imports the reference's own modules behind the shims of oracle/validate_against_reference.py, asserts that the oracle restatement
agrees, and writes the REFERENCE's outputs to tests/golden/reference_multishot.npz.

  python oracle/make_multishot_golden.py
Test infrastructure only."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import validate_against_reference as V  # noqa: E402


NUM_POINTS_CASES = (("default", 3), ("conf_pts", 3), ("cca", 2))     # (flag set of synth_cases.ORCH_FLAGS, num_points_for_sam)


def num_points_records(gold):
    """ProtoSAM.forward with num_points_for_sam = k > 1 (ProtoSAM.py:376-387: the k most confident pixels of every component,
    plus its centroid in 'both' mode) on the orchestration check's inputs: reference vs oracle/glue.py, reference recorded."""
    import tempfile
    import matplotlib
    matplotlib.use("Agg")
    V._install_cv2_restatements()
    from oracle import glue, sam_image_encoder as oenc
    from protosam_amd import synth_cases as gi
    from protosam_amd.synth import synth_state_dict
    V._truncate_vendored_registry(gi.ORCH_SAM_DEPTH)
    import models.ProtoSAM as ref_ps
    from segment_anything import sam_model_registry
    sam_sd = synth_state_dict(sam_model_registry["vit_b"](), gi.ORCH_SAM_SEED)
    D, q, S, logits = gi.ORCH_SAM_DEPTH, gi.orch_query(), gi.ORCH_SIZE, gi.orch_coarse_logits()
    qf = V.F_interp(q)
    feats = oenc.image_encoder(glue.sam_preprocess(glue.quantise_image(qf)), sam_sd, model_type="vit_b", depth=D)
    orig_cpu = torch.Tensor.cpu                       # (shim 5 of validate_against_reference.check_orchestration: .cpu() copies)
    torch.Tensor.cpu = lambda self, *a, **k: orig_cpu(self, *a, **k).clone()
    try:
        with tempfile.TemporaryDirectory() as tmpdir:
            ckpt = os.path.join(tmpdir, "sam_vit_b_synth.pth")
            torch.save(sam_sd, ckpt)
            for name, k in NUM_POINTS_CASES:
                kw = gi.ORCH_FLAGS[name]
                ref_model = ref_ps.ProtoSAM(image_size=(1024, 1024), coarse_segmentation_model=V._FixedCoarse(logits),
                                            sam_pretrained_path=ckpt, num_points_for_sam=k, use_sam_trans=True, **kw).eval()
                inp = ref_ps.InputFactory.create_input(ref_ps.TYPE_ALPNET, q, support_images=[q], support_labels=[torch.zeros(1, S, S)],
                                                       isval=True, val_wsize=2)
                with torch.no_grad():
                    pred_r, scores_r = ref_model(q, inp, degrees_rotate=0)
                    taps = {}
                    pred_o, scores_o = glue.protosam_forward(q, logits, sam_sd, "vit_b", postprocess="batched", encoder_depth=D,
                                                             features=feats, taps=taps, num_points=k, **kw)
                    pred_1, _ = glue.protosam_forward(q, logits, sam_sd, "vit_b", postprocess="batched", encoder_depth=D,
                                                      features=feats, **kw)
                d = int((pred_r != pred_o).sum())
                print(f"  [{'ok' if d == 0 else 'FAIL'}] ProtoSAM.forward {name}, num_points_for_sam={k}: {len(scores_r)} prompt sets, "
                      f"{d} differing pixels, fg {int(pred_r.sum())} ({int((pred_r != pred_1).sum())} pixels away from k = 1)")
                assert d == 0 and len(scores_r) == len(scores_o)
                V.close(np.array(scores_o, dtype=np.float64), np.array([float(v) for v in scores_r]), 1e-5, f"{name} k={k}: scores")
                gold[f"orch_{name}_k{k}_mask"] = V._pack(pred_r.numpy())
                gold[f"orch_{name}_k{k}_scores"] = np.array([float(v) for v in scores_r], dtype=np.float32)
                gold[f"orch_{name}_k{k}_low"] = torch.stack([torch.as_tensor(l) for l in taps["low_res"]])[..., ::4, ::4].numpy().astype(np.float32)
                gold[f"orch_{name}_k{k}_points"] = np.stack([np.asarray(p) for p in taps["points"]]).astype(np.int32)
    finally:
        torch.Tensor.cpu = orig_cpu


def main():
    if not os.path.isdir(V.REF):
        raise SystemExit("/root/reference not present: this script only runs in the build container")
    V.install_shims()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    from oracle import alp as oalp, dinov2 as odino
    from protosam_amd import synth_cases as gi
    gold = {}
    depth = gi.FEWSHOT_DEPTH
    enc_sd = gi.fewshot_encoder_sd()
    torch.hub.load = lambda repo, name, **k: V._HubAdapter("dinov2_b14", enc_sd, depth)
    from models.grid_proto_fewshot import FewShotSeg  # reference
    cfg = {"which_model": "dinov2_b14", "cls_name": "grid_proto", "proto_grid_size": 8, "lora": 0, "align": False,
           "debug": False, "use_coco_init": False}
    enc = lambda im: odino.forward_features(im, enc_sd, "dinov2_b14", depth=depth)["x_norm_patchtokens"]  # noqa
    for size, n_shots in gi.MULTISHOT_CASES:
        ref_model = FewShotSeg(size, None, cfg).eval()
        s_imgs, s_ms, q_img = gi.multishot_inputs(size, n_shots)
        with torch.no_grad():
            ref = ref_model([s_imgs], [s_ms], [[1 - m for m in s_ms]], [q_img], True, 2)[0]
        out = oalp.fewshot_forward_multishot(enc, s_imgs, s_ms, q_img, size)
        V.close(out, ref, 1e-4, f"{n_shots}-shot logits image_size={size}")
        one = oalp.fewshot_forward(enc, s_imgs[0], s_ms[0], q_img, size)
        print(f"    (vs the first shot alone: max |dlogit| {float((ref - one).abs().max()):.3f} - the shots matter)")
        gold[f"fewshot_logits_{size}_{n_shots}shot"] = ref.numpy().astype(np.float32)
    num_points_records(gold)
    path = os.path.join(V.GOLD, "reference_multishot.npz")
    np.savez_compressed(path, **gold)
    print(f"wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB, {len(gold)} arrays)")


if __name__ == "__main__":
    main()
