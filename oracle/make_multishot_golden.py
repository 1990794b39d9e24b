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
    path = os.path.join(V.GOLD, "reference_multishot.npz")
    np.savez_compressed(path, **gold)
    print(f"wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB, {len(gold)} arrays)")


if __name__ == "__main__":
    main()
