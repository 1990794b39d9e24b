"""CPU checks of the rotation-TTA oracle (oracle/rotate.py): the restated torchvision tensor ops against independent
facts (a quarter turn is `rot90`, a null turn is the identity, the expanded canvas bounds the rotated rectangle) and
`reverse_tensor(rotate_tensor_no_crop(x))` restoring the interior of a smooth image."""
import math

import torch

from oracle import rotate as orot


def test_quarter_turns_are_rot90():
    x = torch.rand(2, 3, 40, 40)
    for k, ang in ((1, 90), (2, 180), (3, 270), (-1, -90)):
        y = orot.tv_rotate(x, ang, expand=True)
        assert torch.equal(y, torch.rot90(x, k, dims=(2, 3)))       # positive angles turn counter-clockwise


def test_zero_degrees_is_identity_and_sizes():
    x = torch.rand(1, 3, 33, 47)
    y, sz = orot.rotate_tensor_no_crop(x, 0)
    assert y is x and sz == (33, 47)
    for ang in (10, 37, 45, 80):
        m = orot.inverse_rotation_matrix(-ang)
        ow, oh = orot.affine_output_size(m, 47, 33)
        c, s = abs(math.cos(math.radians(ang))), abs(math.sin(math.radians(ang)))
        assert 0 <= ow - (47 * c + 33 * s) <= 2.0 and 0 <= oh - (47 * s + 33 * c) <= 2.0
        y, (rh, rw) = orot.rotate_tensor_no_crop(x, ang)
        assert y.shape == x.shape and (rh, rw) == (oh, ow)


def test_rotate_then_reverse_restores_the_interior():
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, 128), torch.linspace(-1, 1, 128), indexing="ij")
    img = torch.stack([torch.sin(3 * xx) * torch.cos(2 * yy), xx * yy, xx + yy])[None]
    rot, (rh, rw) = orot.rotate_tensor_no_crop(img, 20)
    back = orot.reverse_tensor(rot, rh, rw, -20)
    assert back.shape == img.shape
    inner = (slice(None), slice(None), slice(32, 96), slice(32, 96))
    assert (back[inner] - img[inner]).abs().max() < 0.08          # two resamplings of a smooth field
