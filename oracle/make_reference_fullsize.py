"""FULL-DEPTH pin of the oracle's SAM path against the reference itself (runs ONLY in the build container, where /root/reference
exists; test infrastructure, see oracle/__init__.py).

  python oracle/make_reference_fullsize.py            # writes tests/golden/reference_fullsize_vith.npz

oracle/validate_against_reference.py pins oracle/sam_image_encoder.py on a reduced-width 3-block ImageEncoderViT and the decoder at
full size; the full-depth records of tests/golden/fullsize_cfg*.npz are ORACLE outputs. This script closes the chain at full depth:
the reference's own `sam_model_registry["vit_h"]` (32 blocks, models/segment_anything/build_sam.py:14-24,
modeling/image_encoder.py:108-122) and its `SamPredictor.set_image / predict` (predictor.py:34-241) run on one slice of BASELINE
config 4 (the benchmark's workload: z = 32 of the seeded CT-like volume, prompts from the coarse stage), and
  * the oracle's image embedding and low-res logits are asserted equal to the reference's (<= 2e-5 / 5e-4 on +-20 logits),
  * the REFERENCE's low-res logits, scores and embedding samples are stored; tests/test_fullsize_gpu.py compares the HIP path
    against them directly (no oracle in between).
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")
Z = 32


def main():
    from oracle.validate_against_reference import REF, close, install_shims
    if not os.path.isdir(REF):
        raise SystemExit("/root/reference not present: this script only runs in the build container")
    install_shims()
    torch.set_num_threads(os.cpu_count() or 1)
    from segment_anything import SamPredictor, sam_model_registry          # the reference's vendored copy
    from oracle import alp as oalp, dinov2 as odino, glue
    from oracle.make_fullsize_goldens import _weights
    from protosam_amd.runner import part_assign, support_set
    from protosam_amd.synth import synth_volume
    from protosam_amd.synth_cases import volume_config
    sam_type, n, kind, _, _ = volume_config(4)
    enc_sd, sam_sd = _weights(sam_type, 512)
    vol, _ = synth_volume(n, 512, seed=0, kind=kind)
    svol, slab = synth_volume(n, 512, seed=1, kind=kind)
    sup_imgs, sup_masks = support_set(svol, slab)
    q = vol[Z][None, None].repeat(1, 3, 1, 1).contiguous()
    part = part_assign(Z, n)
    t0 = time.time()
    with torch.no_grad():
        enc = lambda im: odino.forward_features(im, enc_sd, "dinov2_b14")["x_norm_patchtokens"]  # noqa: E731
        logits = oalp.fewshot_forward(enc, sup_imgs[part], sup_masks[part], q, 512)
        taps = {}
        pred_o, scores_o = glue.protosam_forward(q, logits, sam_sd, sam_type, use_bbox=True, use_points=True, point_mode="both",
                                                 use_cca=False, taps=taps)
    print(f"oracle: {len(scores_o)} component(s), fg {int(pred_o.sum())} px, {time.time() - t0:.0f}s", flush=True)
    # ---- the reference: 32-block ViT-H from its own registry, its own predictor
    t0 = time.time()
    sam = sam_model_registry["vit_h"]()
    missing = sam.load_state_dict(sam_sd, strict=True)
    assert len(sam.image_encoder.blocks) == 32, len(sam.image_encoder.blocks)
    predictor = SamPredictor(sam.eval())
    out = {}
    with torch.no_grad():
        predictor.set_image(taps["img_u8"])                                   # predictor.py:34-88 (1024 x 1024: apply_image is the identity)
        emb_r = predictor.get_image_embedding()
        print(f"reference ImageEncoderViT (ViT-H x32): {time.time() - t0:.0f}s", flush=True)
        close(taps["features"], emb_r, 2e-5, "ViT-H x32 image embedding: oracle vs reference")
        out["embedding_s8"] = emb_r[0, :, ::8, ::8].numpy().astype(np.float32)
        lows, ious, masks = [], [], []
        for i, (point, box) in enumerate(zip(taps["points"], taps["bboxes"])):   # ProtoSAM.py:505-527 with use_cca=False
            labels = np.array([1] * len(point))
            m_r, s_r, low_r = predictor.predict(point_coords=point, point_labels=labels, box=box, multimask_output=True)
            close(taps["low_res"][i], low_r, 5e-4, f"component {i}: low_res logits, oracle vs reference")
            close(scores_o[i], s_r[0], 2e-5, f"component {i}: score of the kept mask")
            lows.append(low_r.astype(np.float32))
            ious.append(s_r.astype(np.float32))
            masks.append(m_r[0])
        # (full-size masks: the vendored predictor post-processes with SamBatched (bilinear, align_corners=True), ProtoSAM.forward
        # imports the pip package's Sam (align_corners=False) - both variants of the oracle are pinned in validate_against_reference.py)
        from oracle import sam_prompt_decoder as odec
        for i, m_r in enumerate(masks):
            m_o = odec.postprocess_masks(torch.from_numpy(lows[i])[None], (1024, 1024), (1024, 1024), "batched")[0, 0] > 0
            d = int((m_o.numpy() != m_r).sum())
            print(f"  [{'ok' if d == 0 else 'FAIL'}] component {i}: full-size mask (SamBatched post-processing): {d} differing pixels")
            assert d == 0
    out["low_res"] = np.stack(lows)                   # [n, 3, 256, 256] fp32 logits of the three mask tokens
    out["iou"] = np.stack(ious)
    out["mask"] = np.packbits(pred_o.numpy().astype(bool))                   # (final mask: union of the kept masks, nearest to 512)
    out["z"] = np.array([Z])
    path = os.path.join(GOLD, "reference_fullsize_vith.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB)\nFULL-DEPTH CHECK PASSED")


if __name__ == "__main__":
    main()
