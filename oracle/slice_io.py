"""Oracle: the reference's host-side slice preparation (numpy). Test infrastructure only (see oracle/__init__.py).

Follows dataloaders/ManualAnnoDatasetv2.py (read_dataset :151-227, __getitem_default__ :317-327) and
dataloaders/dataset_utils.py (MR_normalize :101-102, CT_normalize :104-108).
`cv2.resize` (opencv-python 4.10.0.84) is absent => PARITY UNPINNED; INTER_LINEAR / INTER_NEAREST for float32 images are
restated from their published rule (pixel centres at +0.5, `fx = (dx + 0.5) * scale - 0.5`, floor, edge clamp with the
weight collapsed onto the border pixel, horizontal pass then vertical pass; nearest = floor(dx * scale)) and cross-checked
against torch's `F.interpolate(align_corners=False)` in tests/test_slice_io_cpu.py.
`SimpleITK.ReadImage` is absent as well: `nifti_bytes` builds a NIfTI-1 file image independently of the product's writer
so the product's reader is tested against bytes it did not produce.
"""
import struct

import numpy as np


def resize_linear(img_hw, S):
    """cv2.resize(img, (S, S), interpolation=cv2.INTER_LINEAR) for one float32 [H, W] image."""
    img = np.asarray(img_hw, dtype=np.float32)
    H, W = img.shape

    def taps(n_in):
        f = ((np.arange(S, dtype=np.float64) + 0.5) * (n_in / S) - 0.5).astype(np.float32)
        i0 = np.floor(f).astype(np.int64)
        w = (f - i0.astype(np.float32)).astype(np.float32)
        i1 = i0 + 1
        lo, hi = i0 < 0, i0 >= n_in - 1
        i0[lo], i1[lo], w[lo] = 0, 0, 0
        i0[hi], i1[hi], w[hi] = n_in - 1, n_in - 1, 0
        return i0, i1, w

    x0, x1, wx = taps(W)
    y0, y1, wy = taps(H)
    rows = img[:, x0] * (np.float32(1) - wx)[None, :] + img[:, x1] * wx[None, :]
    return (rows[y0] * (np.float32(1) - wy)[:, None] + rows[y1] * wy[:, None]).astype(np.float32)


def resize_nearest(img_hw, S):
    """cv2.resize(img, (S, S), interpolation=cv2.INTER_NEAREST)."""
    img = np.asarray(img_hw)
    H, W = img.shape
    sx = np.minimum(np.floor(np.arange(S) * (W / S)).astype(np.int64), W - 1)
    sy = np.minimum(np.floor(np.arange(S) * (H / S)).astype(np.int64), H - 1)
    return img[sy][:, sx]


def prepare_scan(vol_zyx, S, modality="MR", ct_mean=None, ct_std=None, tile_z_dim=3, labels_zyx=None):
    """read_dataset + __getitem_default__ for every slice: -> images float32 [Z, tile, S, S], labels float32 [Z, S, S]."""
    img = np.float32(np.asarray(vol_zyx).transpose(1, 2, 0))                     # :167,172
    if modality == "MR":
        img = (img - img.mean()) / img.std()                                    # dataset_utils.py:101-102
    else:
        img = (img - ct_mean) / ct_std                                          # :104-108
    img = np.float32(img)
    out = np.stack([resize_linear(img[..., z], S) for z in range(img.shape[-1])])            # :182 (per channel)
    out = np.repeat(out[:, None], tile_z_dim, axis=1)                                        # :325-327
    lab = None
    if labels_zyx is not None:
        lb = np.float32(np.asarray(labels_zyx).transpose(1, 2, 0))
        lab = np.stack([resize_nearest(lb[..., z], S) for z in range(lb.shape[-1])])         # :183
    return out, lab


def nifti_bytes(vol_zyx, spacing=(1.0, 1.0, 1.0), qoffset=(0.0, 0.0, 0.0), quatern=(0.0, 0.0, 0.0), slope=0.0, inter=0.0,
                endian="<"):
    """A single-file NIfTI-1 image (qform only), built field by field from the specification."""
    a = np.ascontiguousarray(vol_zyx)
    code, bits = {np.dtype(np.int16): (4, 16), np.dtype(np.float32): (16, 32), np.dtype(np.uint8): (2, 8),
                  np.dtype(np.int32): (8, 32)}[a.dtype]
    nz, ny, nx = a.shape
    h = bytearray(352)
    struct.pack_into(endian + "i", h, 0, 348)
    struct.pack_into(endian + "8h", h, 40, 3, nx, ny, nz, 1, 1, 1, 1)
    struct.pack_into(endian + "hh", h, 70, code, bits)
    struct.pack_into(endian + "8f", h, 76, 1.0, spacing[0], spacing[1], spacing[2], 0.0, 0.0, 0.0, 0.0)
    struct.pack_into(endian + "fff", h, 108, 352.0, slope, inter)
    struct.pack_into(endian + "hh", h, 252, 1, 0)
    struct.pack_into(endian + "6f", h, 256, quatern[0], quatern[1], quatern[2], qoffset[0], qoffset[1], qoffset[2])
    h[344:348] = b"n+1\0"
    return bytes(h) + a.astype(a.dtype.newbyteorder(endian)).tobytes()
