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
