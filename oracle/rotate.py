"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the rotation test-time augmentation around the
coarse model - `rotate_tensor_no_crop` (/root/reference/util/utils.py:66-83) and `reverse_tensor` (:40-59), as called from
`ProtoSAM.forward` (/root/reference/models/ProtoSAM.py:544-556).

Both helpers are thin wrappers over torchvision.transforms.functional `rotate` / `resize` on TENSORS. torchvision (pinned
0.15.2, /root/reference/requirements.txt:65) is absent from /root/reference and from this image => PARITY UNPINNED for
those two functions: they are restated here from torchvision 0.15.2's published tensor code path, which itself is plain
torch:
  * `rotate(img, angle, NEAREST, expand, center=None, fill=None)` (functional.py / _functional_tensor.py):
    `_get_inverse_affine_matrix([0, 0], -angle, [0, 0], 1.0, [0, 0])`, output size from `_compute_affine_output_size` when
    expanding, `_gen_affine_grid` (base grid of linspace pixel centres, `bmm` with theta^T / [w/2, h/2]) and
    `grid_sample(mode="nearest", padding_mode="zeros", align_corners=False)`;
  * `resize(img, (h, w), BILINEAR, antialias=True)` = `interpolate(mode="bilinear", align_corners=False, antialias=True)`
    (NEAREST ignores antialias).
The two helpers themselves (canvas bookkeeping, interpolation choice by channel count, resize -> rotate -> crop order) ARE
pinned: oracle/validate_against_reference.py imports /root/reference/util/utils.py with these restated primitives injected
as `torchvision.transforms.functional.rotate / resize` and compares bit for bit (golden vector: tests/golden, `rotate_*`).
The reference's own caller only ever passes degrees_rotate = 0 (validation_protosam.py:388), for which both helpers are the
identity; that case is pinned by the golden vectors of the main path.
"""
import math

import torch
import torch.nn.functional as F


def inverse_rotation_matrix(angle):
    """torchvision `_get_inverse_affine_matrix(center=[0,0], angle, translate=[0,0], scale=1, shear=[0,0])`."""
    rot = math.radians(angle)
    a, b, c, d = math.cos(rot), -math.sin(rot), math.sin(rot), math.cos(rot)
    return [d, -b, 0.0, -c, a, 0.0]


def affine_output_size(matrix, w, h):
    """torchvision `_compute_affine_output_size` -> (ow, oh)."""
    pts = torch.tensor([[-0.5 * w, -0.5 * h, 1.0], [-0.5 * w, 0.5 * h, 1.0], [0.5 * w, 0.5 * h, 1.0],
                        [0.5 * w, -0.5 * h, 1.0]])
    theta = torch.tensor(matrix, dtype=torch.float).view(2, 3)
    new_pts = torch.matmul(pts, theta.T)
    min_vals, _ = new_pts.min(dim=0)
    max_vals, _ = new_pts.max(dim=0)
    min_vals += torch.tensor((w * 0.5, h * 0.5))
    max_vals += torch.tensor((w * 0.5, h * 0.5))
    tol = 1e-4
    cmax = torch.ceil((max_vals / tol).trunc_() * tol)
    cmin = torch.floor((min_vals / tol).trunc_() * tol)
    size = cmax - cmin
    return int(size[0]), int(size[1])


def base_grid_axes(ow, oh):
    """the two linspace vectors of `_gen_affine_grid` (x over the output width, y over the output height)."""
    d = 0.5
    return (torch.linspace(-ow * 0.5 + d, ow * 0.5 + d - 1, steps=ow),
            torch.linspace(-oh * 0.5 + d, oh * 0.5 + d - 1, steps=oh))


def rescaled_theta(matrix, w, h):
    theta = torch.tensor(matrix, dtype=torch.float32).reshape(1, 2, 3)
    return theta.transpose(1, 2) / torch.tensor([0.5 * w, 0.5 * h], dtype=torch.float32)      # [1, 3, 2]


def tv_rotate(img, angle, expand=False):
    """torchvision.transforms.functional.rotate on a float tensor [B, C, H, W], NEAREST, zero fill."""
    matrix = inverse_rotation_matrix(-angle)          # "we need to set -angle" (functional.py rotate)
    h, w = img.shape[-2:]
    ow, oh = affine_output_size(matrix, w, h) if expand else (w, h)
    xg, yg = base_grid_axes(ow, oh)
    base = torch.empty(1, oh, ow, 3, dtype=torch.float32)
    base[..., 0].copy_(xg)
    base[..., 1].copy_(yg.unsqueeze(-1))
    base[..., 2].fill_(1)
    grid = base.view(1, oh * ow, 3).bmm(rescaled_theta(matrix, w, h)).view(1, oh, ow, 2)
    grid = grid.expand(img.shape[0], oh, ow, 2)
    return F.grid_sample(img, grid, mode="nearest", padding_mode="zeros", align_corners=False)


def tv_resize(img, size, nearest=False):
    """torchvision.transforms.functional.resize(img, size, BILINEAR | NEAREST, antialias=True) on a float tensor."""
    if nearest:
        return F.interpolate(img, size=list(size), mode="nearest")
    return F.interpolate(img, size=list(size), mode="bilinear", align_corners=False, antialias=True)


def rotate_tensor_no_crop(image_tensor, degrees):
    """util/utils.py:66-83."""
    if degrees == 0:
        return image_tensor, tuple(image_tensor.shape[-2:])
    b, c, h, w = image_tensor.shape
    rotated = tv_rotate(image_tensor, degrees, expand=True)
    resized = tv_resize(rotated, (h, w), nearest=(c == 1))
    return resized, tuple(rotated.shape[-2:])


def reverse_tensor(tensor, original_h, original_w, degrees):
    """util/utils.py:40-59."""
    _, _, h, w = tensor.shape
    if tuple(tensor.shape[-2:]) != (original_h, original_w):
        tensor = tv_resize(tensor, (original_h, original_w))
    rotated = tv_rotate(tensor, degrees, expand=False)
    h_remove = abs(h - original_h) // 2
    w_remove = abs(w - original_w) // 2
    if h_remove > 0 and w_remove > 0:
        rotated = rotated[:, :, h_remove:-h_remove, w_remove:-w_remove]
    return rotated
