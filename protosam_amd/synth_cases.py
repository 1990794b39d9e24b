"""Seeded inputs shared by oracle/validate_against_reference.py (which records the REFERENCE's outputs for them in
tests/golden/reference_outputs.npz) and tests/test_oracle_golden.py (which replays the oracle on them anywhere)."""
import torch


def alp_case():
    g = torch.Generator().manual_seed(11)
    C, hw = 64, 32
    qry = torch.randn((1, 1, C, hw, hw), generator=g)
    sup = torch.randn((1, 1, 1, C, hw, hw), generator=g)
    msk = torch.zeros((1, 1, 1, hw, hw))
    msk[..., 7:21, 5:23] = 1
    return qry, sup, msk


FEWSHOT_SIZES = (252, 448)
FEWSHOT_DEPTH = 1
FEWSHOT_SEED = 1234


def fewshot_encoder_sd():
    from protosam_amd.dinov2 import DinoVisionTransformer
    from protosam_amd.synth import synth_state_dict
    return synth_state_dict(DinoVisionTransformer("dinov2_vitb14", depth=FEWSHOT_DEPTH), FEWSHOT_SEED)


def fewshot_pair(size):
    from protosam_amd.synth import synth_pair
    return synth_pair(size, seed=size)


MULTISHOT_CASES = ((448, 2), (448, 3), (252, 2))       # (image_size, n_shots) of tests/golden/reference_multishot.npz


def multishot_inputs(size, n_shots):
    """seeded support shots (images, masks) and one query: the pairs of synth.synth_pair with seeds size, size + 1, ..."""
    from protosam_amd.synth import synth_pair
    shots = [synth_pair(size, seed=size + i)[:2] for i in range(n_shots)]
    return [s[0] for s in shots], [s[1] for s in shots], synth_pair(size, seed=size)[2]


SMALL_ENCODER = dict(embed_dim=64, depth=3, num_heads=2, global_attn_indexes=(1,), out_chans=32)
SMALL_ENCODER_SEED = 4321


def small_encoder_input():
    return torch.randn((1, 3, 1024, 1024), generator=torch.Generator().manual_seed(8))


DECODER_SEED = 1234


def decoder_features():
    return torch.randn((1, 256, 64, 64), generator=torch.Generator().manual_seed(21))


def decoder_cases():
    return {
        "pts_box": (torch.tensor([[[300.0, 410.0], [512.5, 600.25]], [[100.0, 90.0], [900.0, 30.5]],
                                  [[5.0, 1000.0], [640.0, 640.0]]]),
                    torch.ones((3, 2), dtype=torch.int),
                    torch.tensor([[250.0, 300.0, 700.0, 800.0], [50.0, 20.0, 950.0, 200.0], [0.0, 600.0, 700.0, 1023.0]])),
        "pts_only": (torch.tensor([[[300.0, 410.0], [20.0, 30.0]]]), torch.tensor([[1, 0]], dtype=torch.int), None),
        "box_only": (None, None, torch.tensor([[250.0, 300.0, 700.0, 800.0]])),
    }


AMG_SEED = 2024
AMG_ENCODER_DEPTH = 2          # vit_b block stack truncated to [window, window] so the CPU reference runs in seconds
AMG_ARGS = dict(points_per_side=8, points_per_batch=32, box_nms_thresh=1.0)
# the crop-layer case on the small image: every candidate kept by the score filters (thresholds 0), holes / islands below 6 px
AMG_CROP_ARGS = dict(points_per_side=8, points_per_batch=32, box_nms_thresh=1.0, pred_iou_thresh=0.0,
                     stability_score_thresh=0.0, crop_n_layers=1, crop_n_points_downscale_factor=2, min_mask_region_area=6)


def amg_case():
    """uint8 [1024,1024,3] image (what SAMWrapperInput hands to SamWrapper.forward) and a binary label [1024,1024]."""
    from protosam_amd.synth import synth_pair
    _, _, q_img, q_gt = synth_pair(1024, seed=5)
    q = q_img[0].permute(1, 2, 0).numpy()
    img = ((q - q.min()) / (q.max() - q.min()) * 255).astype("uint8")
    return img, q_gt[0].numpy().astype("uint8")


def mask_prompt_case():
    """[2,1,256,256] float masks with the values ProtoSAM.predict_w_masks produces (10 inside, uint8(-8) = 248 outside)."""
    m = torch.full((2, 1, 256, 256), 248.0)
    m[0, 0, 60:140, 90:200] = 10.0
    yy, xx = torch.meshgrid(torch.arange(256.0), torch.arange(256.0), indexing="ij")
    m[1, 0][((yy - 150) / 40) ** 2 + ((xx - 100) / 70) ** 2 <= 1] = 10.0
    return m


# ---- orchestration cases: reference ProtoSAM.forward / ProtoMedSAM.forward / SamPredictor run end to end ----------------
ORCH_SAM_DEPTH = 2            # ViT-B block stack truncated to [window, window] so the CPU reference runs in seconds
ORCH_SAM_SEED = 1234
ORCH_SIZE = 512
# flag sets of ProtoSAM.__init__ the reference is run with (validation_protosam.py:220-232 passes exactly these names)
ORCH_FLAGS = {
    "default": dict(use_bbox=True, use_points=True, point_mode="both", use_cca=False),
    "cca": dict(use_bbox=True, use_points=True, point_mode="both", use_cca=True),
    "conf_pts": dict(use_bbox=False, use_points=True, point_mode="conf", use_cca=False),
    "centroid_box": dict(use_bbox=True, use_points=True, point_mode="centroid", use_cca=False),
    "box_only": dict(use_bbox=True, use_points=False, use_cca=False),
    "mask": dict(use_bbox=False, use_points=False, use_mask=True, use_cca=False),
    "mask_cca": dict(use_bbox=False, use_points=False, use_mask=True, use_cca=True),
    "neg": dict(use_bbox=True, use_points=True, point_mode="both", use_cca=False, use_neg_points=True),
}


def orch_query():
    """The query slice [1,3,512,512] every orchestration case uses (one SAM encoding serves all of them)."""
    return fewshot_pair(ORCH_SIZE)[2]


def orch_coarse_logits(seed=3, size=ORCH_SIZE, thr=1.0, gain=6.0):
    """[1,2,S,S] stand-in for the coarse model's logits: a smooth random field with several separate foreground blobs
    (one of them tiny), scaled like ALP logits so that softmax probabilities cover (0,1) around the blob borders."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(seed)
    f = torch.randn((1, 1, size // 64, size // 64), generator=g)
    f = F.interpolate(f, size=(size, size), mode="bicubic", align_corners=False)[0, 0]
    f = (f - f.mean()) / f.std()
    # a 3x4-pixel component of its own; values all different: on a plateau `torch.topk` (ProtoSAM.py:281) may return any
    # of the tied pixels (CPU and CUDA differ), while the oracle and the HIP path take the first one in raster order
    # (and kept below the level where softmax saturates to exactly 1.0f - logit difference > ~16.6 - for the same reason)
    f[40:43, 60:64] = 1.7 + 0.01 * torch.arange(12.0).reshape(3, 4).flip(1)
    # bounded (|logit difference| <= 2 * gain) so that no probability saturates to exactly 1.0f: see the tie note above
    fg = gain * torch.tanh((f - thr) / 1.5)
    return torch.stack([-fg, fg])[None].contiguous()


def orch_empty_logits(size=ORCH_SIZE):
    z = torch.zeros((1, 2, size, size))
    z[:, 0] = 5.0
    return z


def predictor_cases():
    """(name, image HxW, point_coords, point_labels, box, mask_input?, multimask_output, return_logits)"""
    import numpy as np
    return [
        ("sq_pts_box", (1024, 1024), np.array([[400.0, 500.0], [520.5, 480.0]]), np.array([1, 1]),
         np.array([300, 350, 700, 800]), False, True, False),
        ("sq_box_single", (1024, 1024), None, None, np.array([300, 350, 700, 800]), False, False, True),
        ("rect_pts_neg", (600, 900), np.array([[360.0, 300.0], [540.0, 180.0]]), np.array([1, 0]), None, False, True, False),
        ("sq_mask_in", (1024, 1024), None, None, None, True, True, False),
    ]


def predictor_image(hw, seed=0):
    """uint8 HWC image of size hw."""
    import numpy as np
    g = torch.Generator().manual_seed(100 + seed + hw[0])
    f = torch.nn.functional.interpolate(torch.randn((1, 3, hw[0] // 32 + 1, hw[1] // 32 + 1), generator=g), size=hw,
                                        mode="bilinear")[0]
    f = (f - f.min()) / (f.max() - f.min()) * 255
    return f.permute(1, 2, 0).numpy().astype(np.uint8)


def amg_small_case():
    """uint8 [48,44,3]: small enough that every crop edge is within is_box_near_crop_edge's 20-pixel tolerance of the image
    edge (utils/amg.py:78-88; layer-1 crops are 32 x 30 at offsets 0 / 17 and 0 / 15), so masks of the layer-1 crops survive
    that filter even with noise-like synthetic masks; it also makes set_image resize every crop (to a long side of 1024) and
    postprocess_masks take its second resize, on non-square crops."""
    from protosam_amd.synth import synth_pair
    _, _, q_img, _ = synth_pair(64, seed=9)
    q = q_img[0, :, 8:56, 10:54].permute(1, 2, 0).numpy()
    return ((q - q.min()) / (q.max() - q.min()) * 255).astype("uint8")


# ---- full-depth cases (tests/golden/fullsize_cfg*.npz, fullvolume_cfg*.npz; BASELINE.json configs 3 / 4 / 5)
CFG3_SLICES = (5, 16, 27)
CFG4_SLICES = (8, 32, 56)
CFG5_SEED = 2


def volume_config(cfg):
    """-> (sam_type, n_slices, kind, slices, flag sets) of configs 3 / 4."""
    if cfg == 3:
        return "vit_b", 32, "mri", CFG3_SLICES, {"default": dict(use_cca=False), "cca": dict(use_cca=True)}
    if cfg == 44:   # config 4 with the heavy-tailed SAM weights of synth.heavy_tail_sam_ (stress case of the fp16 operand path)
        return "vit_h", 64, "ct", (32,), {"default": dict(use_cca=False)}
    return "vit_h", 64, "ct", CFG4_SLICES, {"default": dict(use_cca=False)}


def cfg5_inputs():
    """support image, the four classes' support masks, query image (protosam_amd.synth.synth_pair_multi: four organs of
    different contrast in one 1024 x 1024 slice)."""
    from protosam_amd.synth import synth_pair_multi
    s_img, s_masks, q_img, _ = synth_pair_multi(1024, seed=CFG5_SEED)
    return s_img, s_masks, q_img


# ---- stand-alone module forwards (tests/golden/reference_modules.npz, oracle/make_module_goldens.py) --------------------
MODULE_SEED = 606
MODULE_DIM, MODULE_HEADS = 768, 12          # SAM ViT-B's block width (head size 64)


def _randn(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


def module_block_input():
    """[1,64,64,768]: what `ImageEncoderViT` hands to a block (image_encoder.py:112-113)."""
    return _randn((1, 64, 64, MODULE_DIM), 31)


def module_mlp_input():
    return _randn((300, MODULE_DIM), 32)


def module_ln2d_input():
    return _randn((2, 256, 16, 16), 33, 2.0) + 0.5


def module_patch_input():
    return _randn((1, 3, 1024, 1024), 34)


def module_transformer_inputs():
    """image_embedding [2,256,64,64], image_pe [2,256,64,64] (one grid repeated, mask_decoder.py:128), point_embedding [2,7,256]"""
    pe = _randn((1, 256, 64, 64), 36).expand(2, -1, -1, -1).contiguous()
    return _randn((2, 256, 64, 64), 35), pe, _randn((2, 7, 256), 37)


def module_sam_forward_input():
    """`Sam.forward`'s batched_input (sam.py:54-131): a square image with points + boxes, a 768 x 1024 one (zero padding, un-padded at
    'image_size') with points only; both with the 'image_size' key `SamBatched.forward` reads (sam.py:283)."""
    a = torch.from_numpy(predictor_image((1024, 1024), seed=3)).permute(2, 0, 1).float().contiguous()
    b = torch.from_numpy(predictor_image((768, 1024), seed=4)).permute(2, 0, 1).float().contiguous()
    return [
        dict(image=a, original_size=(512, 512), image_size=(1024, 1024),
             point_coords=torch.tensor([[[400.0, 500.0], [520.5, 480.0]], [[200.0, 700.0], [100.0, 90.0]]]),
             point_labels=torch.tensor([[1, 1], [1, 0]], dtype=torch.int),
             boxes=torch.tensor([[300.0, 350.0, 700.0, 800.0], [50.0, 600.0, 400.0, 900.0]])),
        dict(image=b, original_size=(384, 512), image_size=(768, 1024),
             point_coords=torch.tensor([[[360.0, 300.0], [540.0, 180.0]]]), point_labels=torch.tensor([[1, 0]], dtype=torch.int)),
    ]


def module_segment_all_inputs():
    """query image [1,3,1024,1024] and a label [1,1024,1024] for `ProtoMedSAM.segment_all` (ProtoMedSAM.py:224-249)."""
    from protosam_amd.synth import synth_pair
    _, _, q_img, q_gt = synth_pair(1024, seed=7)
    return q_img, q_gt.reshape(1, 1024, 1024).float()
