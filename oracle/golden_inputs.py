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
