"""`ResizeLongestSide` (models/segment_anything/utils/transforms.py:17-148): coordinate / box scaling, the target shape
rule and the numpy image resize. On the hot path every image is already 1024x1024 and resizing is the identity; other
sizes go through PIL's bilinear resize on the host exactly as the reference does (`resize(to_pil_image(image), size)`
is `PIL.Image.resize(size[::-1], BILINEAR)`, :38). The antialiased torch resize of `apply_image_torch` is only ever the
identity for ProtoSAM and raises otherwise."""
from copy import deepcopy

import numpy as np
import torch


class ResizeLongestSide:
    def __init__(self, target_length, pixel_mean=(123.675, 116.28, 103.53), pixel_std=(58.395, 57.12, 57.375)):
        self.target_length = target_length
        self.pixel_mean = torch.Tensor(list(pixel_mean)).view(-1, 1, 1)
        self.pixel_std = torch.Tensor(list(pixel_std)).view(-1, 1, 1)

    @staticmethod
    def get_preprocess_shape(oldh, oldw, long_side_length):
        scale = long_side_length * 1.0 / max(oldh, oldw)
        newh, neww = oldh * scale, oldw * scale
        return (int(newh + 0.5), int(neww + 0.5))

    def apply_image(self, image):
        target = self.get_preprocess_shape(image.shape[0], image.shape[1], self.target_length)
        if target == tuple(image.shape[:2]):
            return np.array(image)
        from PIL import Image  # host-side, as in the reference (torchvision's PIL path)
        return np.array(Image.fromarray(image).resize((target[1], target[0]), Image.BILINEAR))

    def apply_image_torch(self, image):
        if self.get_preprocess_shape(image.shape[-2], image.shape[-1], self.target_length) != tuple(image.shape[-2:]):
            raise NotImplementedError("antialiased resize of non-1024 images is outside the accelerated path")
        return image

    def apply_coords(self, coords, original_size):
        old_h, old_w = original_size
        new_h, new_w = self.get_preprocess_shape(original_size[0], original_size[1], self.target_length)
        coords = deepcopy(coords).astype(float)
        coords[..., 0] = coords[..., 0] * (new_w / old_w)
        coords[..., 1] = coords[..., 1] * (new_h / old_h)
        return coords

    def apply_boxes(self, boxes, original_size):
        return self.apply_coords(boxes.reshape(-1, 2, 2), original_size).reshape(-1, 4)

    def apply_coords_torch(self, coords, original_size):
        old_h, old_w = original_size
        new_h, new_w = self.get_preprocess_shape(original_size[0], original_size[1], self.target_length)
        coords = deepcopy(coords).to(torch.float)
        coords[..., 0] = coords[..., 0] * (new_w / old_w)
        coords[..., 1] = coords[..., 1] * (new_h / old_h)
        return coords

    def apply_boxes_torch(self, boxes, original_size):
        return self.apply_coords_torch(boxes.reshape(-1, 2, 2), original_size).reshape(-1, 4)

    def preprocess(self, x):
        """Normalise + pad (transforms.py:94-110); with ProtoSAM's mean 0 / std 1 override this is the identity."""
        if len(x.shape) != 2:
            dev = x.device
            x = (x - self.pixel_mean.to(dev)) / self.pixel_std.to(dev)
        h, w = x.shape[-2:]
        if (h, w) != (self.target_length, self.target_length):
            raise NotImplementedError("padding of non-square inputs is outside the accelerated path")
        return x
