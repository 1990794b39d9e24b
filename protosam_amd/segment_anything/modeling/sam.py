"""`Sam` container (names of models/segment_anything/modeling/sam.py:18-173,176-333).

Three `postprocess_masks` conventions exist for the reference: the pip package segment_anything 1.0
that `models/ProtoSAM.py:8` actually imports (bilinear, align_corners=False) - our default; the vendored `SamBatched`
that the vendored registry builds (sam.py:313-320: bilinear, align_corners=True) and the vendored `Sam` (sam.py:154-160:
nearest). Select with `Sam.postprocess_variant` in {"upstream", "batched", "nearest"}.
"""
import torch
import torch.nn as nn

from ... import ops

_VARIANT = {"upstream": 0, "batched": 1, "nearest": 2}


class Sam(nn.Module):
    mask_threshold: float = 0.0
    image_format: str = "RGB"
    postprocess_variant: str = "upstream"

    def __init__(self, image_encoder, prompt_encoder, mask_decoder, pixel_mean=(123.675, 116.28, 103.53),
                 pixel_std=(58.395, 57.12, 57.375)):
        super().__init__()
        self.image_encoder = image_encoder
        self.prompt_encoder = prompt_encoder
        self.mask_decoder = mask_decoder
        self.register_buffer("pixel_mean", torch.Tensor(list(pixel_mean)).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.Tensor(list(pixel_std)).view(-1, 1, 1), False)
        self._mean_host = tuple(float(v) for v in pixel_mean)
        self._std_host = tuple(float(v) for v in pixel_std)

    @property
    def device(self):
        return self.pixel_mean.device

    def variant_id(self):
        return _VARIANT[self.postprocess_variant]

    def preprocess(self, x):
        """sam.py:163-173: (x - pixel_mean) / pixel_std, then zero-pad bottom / right to the square model input."""
        h, w = x.shape[-2:]
        S = self.image_encoder.img_size
        if h > S or w > S:
            raise ValueError(f"preprocess expects images with long side <= {S}, got {(h, w)}")
        if x.dtype not in (torch.uint8, torch.float32):
            x = x.float()
        if x.dim() == 3:                                     # one [3,H,W] image (Sam.forward, sam.py:98)
            return self.preprocess(x[None])[0]
        y = ops.normalize_chw(x.contiguous(), self._mean_host, self._std_host)
        if (h, w) == (S, S):
            return y
        out = torch.zeros(tuple(y.shape[:-2]) + (S, S), dtype=torch.float32, device=y.device)
        out[..., :h, :w] = y
        return out

    def postprocess_masks(self, masks, input_size, original_size):
        """low-res logits [B,C,256,256] -> [B,C,*original_size]: up-sample to the model input, drop the padding, resize to
        the original image (sam.py:132-160 / :296-320)."""
        S = self.image_encoder.img_size
        ih, iw = (int(v) for v in input_size)
        oh, ow = (int(v) for v in original_size)
        up = ops.mask_upsample(masks.float().contiguous(), S, self.variant_id())
        if (ih, iw) != (S, S):
            up = up[..., :ih, :iw].contiguous()
        if (oh, ow) == (ih, iw):
            return up
        return ops.resize2d(up, oh, ow, self.variant_id())

    @torch.no_grad()
    def forward(self, batched_input, multimask_output):
        """sam.py:54-131 (`Sam.forward`) / :212-290 (`SamBatched.forward`): a list over images of dicts with 'image' ([3,H,W], already
        resized to the model's input frame), 'original_size', and optionally 'point_coords' [B,N,2] + 'point_labels' [B,N], 'boxes'
        [B,4], 'mask_inputs' [B,1,256,256] (all in the input frame) -> a list of dicts with 'masks' (bool [B,C,*original_size]),
        'iou_predictions' [B,C], 'low_res_logits' [B,C,256,256]. The padding is removed at 'image_size' where the record carries it
        (SamBatched, :283) and at the image's own size otherwise (Sam, :122). One image-encoder call for all images (the GEMMs see
        every image's tokens), then prompt encoder + two-way decoder + `postprocess_masks` per image."""
        input_images = torch.stack([self.preprocess(x["image"].to(self.device)) for x in batched_input], dim=0)
        tokens = self.image_encoder.forward_tokens(input_images)                       # [n_img, 4096, 256], token-major
        g = self.image_encoder.grid
        outputs = []
        for i, rec in enumerate(batched_input):
            points = (rec["point_coords"], rec["point_labels"]) if "point_coords" in rec else None
            sparse, dense = self.prompt_encoder(points=points, boxes=rec.get("boxes", None), masks=rec.get("mask_inputs", None))
            emb = tokens[i].view(1, g, g, -1).permute(0, 3, 1, 2)
            low_res, iou = self.mask_decoder(image_embeddings=emb, image_pe=self.prompt_encoder.get_dense_pe(),
                                             sparse_prompt_embeddings=sparse, dense_prompt_embeddings=dense,
                                             multimask_output=multimask_output)
            low_res, iou = low_res.contiguous().clone(), iou.clone()                    # (views of the decoder's workspace)
            input_size = rec["image_size"] if "image_size" in rec else tuple(rec["image"].shape[-2:])
            masks = self.postprocess_masks(low_res, input_size=input_size, original_size=rec["original_size"])
            outputs.append({"masks": masks > self.mask_threshold, "iou_predictions": iou, "low_res_logits": low_res})
        return outputs


class SamBatched(Sam):
    """The class the vendored registry builds (build_sam.py:66, sam.py:176-333): `postprocess_masks` with align_corners=True, and
    `forward` reads the un-padded size from the record's 'image_size' (sam.py:283)."""
    postprocess_variant: str = "batched"

    def forward(self, batched_input, multimask_output):
        for rec in batched_input:
            if "image_size" not in rec:
                raise KeyError("image_size")                                            # sam.py:283 indexes it
        return super().forward(batched_input, multimask_output)
