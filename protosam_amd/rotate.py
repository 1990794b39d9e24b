"""Rotation test-time augmentation around the coarse model: `rotate_tensor_no_crop` / `reverse_tensor` with the names,
arguments and results of /root/reference/util/utils.py:66-83 and :40-59 (called from models/ProtoSAM.py:544-556), on the
device. The per-pixel work is two HIP kernels (`psam_rotate_nearest`, `psam_resize_aa`); the host computes what torchvision
computes on the host: the 2x3 inverse matrix, the expanded canvas size, the base-grid axes and the rescaled theta
(a handful of scalars and two short vectors per call).

torchvision 0.15.2 is not part of /root/reference: the restated algorithm is documented in oracle/rotate.py
(parity unpinned for these two helpers; degrees == 0, the only value the reference's caller passes, is the identity)."""
import math

import torch

from . import ops


def _inverse_rotation_matrix(angle):
    rot = math.radians(angle)
    a, b, c, d = math.cos(rot), -math.sin(rot), math.sin(rot), math.cos(rot)
    return [d, -b, 0.0, -c, a, 0.0]


def _affine_output_size(matrix, w, h):
    pts = torch.tensor([[-0.5 * w, -0.5 * h, 1.0], [-0.5 * w, 0.5 * h, 1.0], [0.5 * w, 0.5 * h, 1.0],
                        [0.5 * w, -0.5 * h, 1.0]])
    theta = torch.tensor(matrix, dtype=torch.float).view(2, 3)
    new_pts = torch.matmul(pts, theta.T)
    lo = new_pts.min(dim=0)[0] + torch.tensor((w * 0.5, h * 0.5))
    hi = new_pts.max(dim=0)[0] + torch.tensor((w * 0.5, h * 0.5))
    tol = 1e-4
    size = torch.ceil((hi / tol).trunc_() * tol) - torch.floor((lo / tol).trunc_() * tol)
    return int(size[0]), int(size[1])


def _rotate(img, angle, expand, crop=(0, 0)):
    """torchvision rotate (NEAREST, zero fill) of a float tensor [B,C,H,W]; `crop` = rows / columns removed on each side."""
    matrix = _inverse_rotation_matrix(-angle)
    B, C, h, w = img.shape
    ow, oh = _affine_output_size(matrix, w, h) if expand else (w, h)
    d = 0.5
    xg = torch.linspace(-ow * 0.5 + d, ow * 0.5 + d - 1, steps=ow)
    yg = torch.linspace(-oh * 0.5 + d, oh * 0.5 + d - 1, steps=oh)
    rt = (torch.tensor(matrix, dtype=torch.float32).reshape(2, 3).t()
          / torch.tensor([0.5 * w, 0.5 * h], dtype=torch.float32)).contiguous()           # [3, 2]
    cy, cx = crop
    return ops.rotate_nearest(img, xg.to(img.device), yg.to(img.device), rt, cy, cx, oh - 2 * cy, ow - 2 * cx)


def rotate_tensor_no_crop(image_tensor, degrees):
    if degrees == 0:
        return image_tensor, tuple(image_tensor.shape[-2:])
    b, c, h, w = image_tensor.shape
    if c == 1:
        raise NotImplementedError("single-channel (NEAREST resize) inputs are not on the HIP path")
    rotated = _rotate(image_tensor.float().contiguous(), degrees, expand=True)
    return ops.resize_aa(rotated, h, w), tuple(rotated.shape[-2:])


def reverse_tensor(tensor, original_h, original_w, degrees):
    _, _, h, w = tensor.shape
    t = tensor.float().contiguous()
    if (h, w) != (original_h, original_w):
        t = ops.resize_aa(t, original_h, original_w)
    h_remove = abs(h - original_h) // 2
    w_remove = abs(w - original_w) // 2
    crop = (h_remove, w_remove) if (h_remove > 0 and w_remove > 0) else (0, 0)
    return _rotate(t, degrees, expand=False, crop=crop)
