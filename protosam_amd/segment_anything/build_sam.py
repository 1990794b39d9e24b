"""`sam_model_registry` (models/segment_anything/build_sam.py:14-107) over the HIP-backed modules."""
from functools import partial

import torch

from .modeling import ImageEncoderViT, MaskDecoder, PromptEncoder, Sam, TwoWayTransformer


def build_sam_vit_h(checkpoint=None, **kw):
    return _build_sam(1280, 32, 16, [7, 15, 23, 31], checkpoint, **kw)


build_sam = build_sam_vit_h


def build_sam_vit_l(checkpoint=None, **kw):
    return _build_sam(1024, 24, 16, [5, 11, 17, 23], checkpoint, **kw)


def build_sam_vit_b(checkpoint=None, **kw):
    return _build_sam(768, 12, 12, [2, 5, 8, 11], checkpoint, **kw)


sam_model_registry = {"default": build_sam_vit_h, "vit_h": build_sam_vit_h, "vit_l": build_sam_vit_l,
                      "vit_b": build_sam_vit_b}


def _build_sam(encoder_embed_dim, full_depth, encoder_num_heads, encoder_global_attn_indexes, checkpoint=None,
               encoder_depth=None):
    """encoder_depth (test hook): truncate the block stack; only meaningful with random-weight fixtures."""
    prompt_embed_dim, image_size, vit_patch_size = 256, 1024, 16
    image_embedding_size = image_size // vit_patch_size
    depth = full_depth if encoder_depth is None else encoder_depth
    sam = Sam(
        image_encoder=ImageEncoderViT(depth=depth, embed_dim=encoder_embed_dim, img_size=image_size, mlp_ratio=4,
                                      norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), num_heads=encoder_num_heads,
                                      patch_size=vit_patch_size, qkv_bias=True, use_rel_pos=True,
                                      global_attn_indexes=encoder_global_attn_indexes, window_size=14,
                                      out_chans=prompt_embed_dim),
        prompt_encoder=PromptEncoder(embed_dim=prompt_embed_dim,
                                     image_embedding_size=(image_embedding_size, image_embedding_size),
                                     input_image_size=(image_size, image_size), mask_in_chans=16),
        mask_decoder=MaskDecoder(num_multimask_outputs=3,
                                 transformer=TwoWayTransformer(depth=2, embedding_dim=prompt_embed_dim, mlp_dim=2048,
                                                               num_heads=8),
                                 transformer_dim=prompt_embed_dim, iou_head_depth=3, iou_head_hidden_dim=256),
        pixel_mean=[123.675, 116.28, 103.53], pixel_std=[58.395, 57.12, 57.375])
    sam.eval()
    if checkpoint is not None:
        with open(checkpoint, "rb") as f:
            state_dict = torch.load(f)
        sam.load_state_dict(state_dict)
    return sam
