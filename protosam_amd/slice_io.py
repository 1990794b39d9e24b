"""Slice I/O either side of the hot path: scan volume file -> normalised, resized, z-tiled slices on the
device, and predicted mask volume -> file.

Mirrors, for inference, what the reference does on the host:
  * `read_nii_bysitk` / `convert_to_sitk` + `sitk.WriteImage` (dataloaders/niftiio.py:10-36, validation.py:325-329),
  * `read_dataset` (dataloaders/ManualAnnoDatasetv2.py:151-227): float32 -> `norm_func` -> `cv2.resize(INTER_LINEAR)` of the
    image, `cv2.resize(INTER_NEAREST)` of the label, and `__getitem_default__` (:317-327): `tile_z_dim` channel repeat,
  * `MR_normalize` / `CT_normalize` / `get_CT_statistics` (dataloaders/dataset_utils.py:76-108).

The file format work (NIfTI-1 header, gzip) stays on the host; the voxels are uploaded ONCE in their stored type and every
per-voxel step (scaling, statistics, normalisation, both resizes, tiling) runs in two HIP kernels (`psam_volume_stats`,
`psam_volume_slices`), so a scan streams into `run_slices` without a float32 host copy or per-slice host work.

SimpleITK is absent here and from /root/reference => PARITY UNPINNED for the file reader; it is restated from the NIfTI-1
specification (348-byte header, `dim`, `datatype`, `pixdim`, `vox_offset`, `scl_slope/inter`, qform / sform). Like ITK's
reader, geometry is reported in LPS (x and y flipped from NIfTI's RAS) and `scl_slope != 0` rescales the voxels.
"""
import gzip
import os
import struct

import numpy as np
import torch

from . import ops

_DTYPES = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64, 256: np.int8, 512: np.uint16,
           768: np.uint32}
_CODES = {np.dtype(np.uint8): (2, 8), np.dtype(np.int16): (4, 16), np.dtype(np.int32): (8, 32),
          np.dtype(np.float32): (16, 32)}
_DEVICE_DT = {np.dtype(np.int16): 0, np.dtype(np.float32): 1, np.dtype(np.uint8): 2, np.dtype(np.int32): 3}
_TORCH_DT = {torch.int16: 0, torch.float32: 1, torch.uint8: 2, torch.int32: 3}


def _open(path, mode):
    return gzip.open(path, mode) if str(path).endswith(".gz") else open(path, mode)


def read_nifti(path, peel_info=False):
    """-> voxels [z, y, x] in their stored type (rescaled to float32 if the header carries a slope), optionally with
    {"spacing", "origin", "direction", "array_size"} as `read_nii_bysitk(peel_info=True)` reports them, plus the raw
    header bytes and (slope, inter)."""
    with _open(path, "rb") as f:
        raw = f.read()
    if len(raw) < 352:
        raise ValueError(f"{path}: too short for a NIfTI-1 file")
    end = "<"
    if struct.unpack("<i", raw[0:4])[0] != 348:
        end = ">"
        if struct.unpack(">i", raw[0:4])[0] != 348:
            raise ValueError(f"{path}: not a NIfTI-1 header (sizeof_hdr != 348)")
    if raw[344:348] not in (b"n+1\0", b"ni1\0"):
        raise ValueError(f"{path}: bad NIfTI magic {raw[344:348]!r}")
    dim = struct.unpack(end + "8h", raw[40:56])
    datatype, bitpix = struct.unpack(end + "2h", raw[70:74])
    pixdim = struct.unpack(end + "8f", raw[76:108])
    vox_offset, slope, inter = struct.unpack(end + "3f", raw[108:120])
    if datatype not in _DTYPES:
        raise ValueError(f"{path}: unsupported NIfTI datatype {datatype}")
    nd = dim[0]
    if nd < 3 or any(d != 1 for d in dim[4:nd + 1]):
        raise ValueError(f"{path}: expected a 3-D volume, dim = {dim}")
    nx, ny, nz = dim[1:4]
    dt = np.dtype(_DTYPES[datatype]).newbyteorder(end)
    if raw[344:348] == b"n+1\0":
        data, off = raw, int(vox_offset)
    else:   # "ni1": the two-file form, voxels live in the sibling .img (offset vox_offset there, normally 0)
        img = None
        for a, b in ((".hdr.gz", ".img.gz"), (".hdr", ".img")):
            if str(path).endswith(a):
                img = str(path)[:-len(a)] + b
                break
        if img is None or not os.path.exists(img):
            raise ValueError(f"{path}: two-file NIfTI (magic 'ni1') without a readable .img beside it")
        with _open(img, "rb") as f:
            data = f.read()
        off = int(vox_offset)
    need = off + nx * ny * nz * dt.itemsize
    if off < 0 or len(data) < need:
        raise ValueError(f"{path}: voxel data truncated ({len(data)} bytes, header promises {need})")
    vox = np.frombuffer(data, dtype=dt, count=nx * ny * nz, offset=off).reshape(nz, ny, nx)
    vox = vox.astype(dt.newbyteorder("="), copy=False)
    scaled = slope != 0.0 and not (slope == 1.0 and inter == 0.0)
    if scaled:
        vox = vox.astype(np.float32) * np.float32(slope) + np.float32(inter)
    if not peel_info:
        return vox
    qform_code, sform_code = struct.unpack(end + "2h", raw[252:256])
    spacing = tuple(float(v) for v in pixdim[1:4])
    if qform_code > 0:
        b, c, d, qx, qy, qz = struct.unpack(end + "6f", raw[256:280])
        a = np.sqrt(max(0.0, 1.0 - (b * b + c * c + d * d)))
        R = np.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
                      [2 * (b * c + a * d), a * a + c * c - b * b - d * d, 2 * (c * d - a * b)],
                      [2 * (b * d - a * c), 2 * (c * d + a * b), a * a + d * d - b * b - c * c]], dtype=np.float64)
        if pixdim[0] < 0:
            R[:, 2] *= -1
        org = np.array([qx, qy, qz], dtype=np.float64)
    elif sform_code > 0:
        M = np.array(struct.unpack(end + "12f", raw[280:328]), dtype=np.float64).reshape(3, 4)
        R = M[:, :3] / np.maximum(np.linalg.norm(M[:, :3], axis=0, keepdims=True), 1e-30)
        org = M[:, 3].copy()
    else:
        R, org = np.eye(3), np.zeros(3)
    lps = np.diag([-1.0, -1.0, 1.0])                                            # NIfTI RAS -> ITK LPS
    info = {"spacing": spacing, "origin": tuple(float(v) for v in lps @ org),
            "direction": tuple(float(v) for v in (lps @ R).reshape(-1)), "array_size": vox.shape,
            "header": bytes(raw[:348]), "scaling": (float(slope), float(inter)) if scaled else (1.0, 0.0)}
    return vox, info


def write_nifti(path, array_zyx, peeled_info=None):
    """`sitk.WriteImage(convert_to_sitk(array, peeled_info), path, True)`: [z, y, x] array -> NIfTI-1 (.nii / .nii.gz) with
    the spacing / origin / direction of `peeled_info` (as returned by `read_nifti(peel_info=True)`), identity otherwise."""
    a = np.ascontiguousarray(array_zyx)
    if a.dtype == np.bool_:
        a = a.astype(np.uint8)
    if a.dtype == np.float64:
        a = a.astype(np.float32)
    if a.dtype not in _CODES or a.ndim != 3:
        raise ValueError(f"write_nifti: unsupported array {a.dtype} {a.shape}")
    code, bits = _CODES[a.dtype]
    nz, ny, nx = a.shape
    spacing = (1.0, 1.0, 1.0)
    R_lps, org_lps = np.eye(3), np.zeros(3)                                     # sitk.GetImageFromArray's defaults
    if peeled_info:
        spacing = tuple(float(v) for v in peeled_info["spacing"])
        R_lps = np.asarray(peeled_info["direction"], dtype=np.float64).reshape(3, 3)
        org_lps = np.asarray(peeled_info["origin"], dtype=np.float64)
    lps = np.diag([-1.0, -1.0, 1.0])                                            # ITK LPS -> NIfTI RAS
    R, org = lps @ R_lps, lps @ org_lps
    M = R * np.asarray(spacing)[None, :]
    hdr = bytearray(348)
    struct.pack_into("<i", hdr, 0, 348)
    struct.pack_into("<8h", hdr, 40, 3, nx, ny, nz, 1, 1, 1, 1)
    struct.pack_into("<2h", hdr, 70, code, bits)
    struct.pack_into("<8f", hdr, 76, 1.0, spacing[0], spacing[1], spacing[2], 1.0, 1.0, 1.0, 1.0)
    struct.pack_into("<3f", hdr, 108, 352.0, 1.0, 0.0)
    hdr[123] = 2                                                                # xyzt_units: millimetres
    struct.pack_into("<2h", hdr, 252, 0, 1)                                     # sform only (exact for any direction)
    struct.pack_into("<12f", hdr, 280, *np.concatenate([M, org[:, None]], axis=1).reshape(-1))
    hdr[344:348] = b"n+1\0"
    with _open(path, "wb") as f:
        f.write(bytes(hdr) + b"\0\0\0\0" + a.tobytes())


class ScanSlices:
    """A scan on the device, ready for the per-slice loop: `images` fp32 [Z, tile, S, S] and (optionally) `labels` fp32
    [Z, S, S], built from the raw voxel arrays by the two HIP kernels."""

    def __init__(self, images, labels, mean, std, info=None):
        self.images, self.labels, self.mean, self.std, self.info = images, labels, mean, std, info

    @staticmethod
    def volume_stats(vol_zyx, device, scaling=(1.0, 0.0)):
        """(mean, std) of the whole volume in fp64 (`x.mean()`, `x.std()` of MR_normalize; get_CT_statistics per scan)."""
        v = ScanSlices._upload(vol_zyx, device)
        s = ops.volume_stats(v, _TORCH_DT[v.dtype], scaling[0], scaling[1]).cpu().numpy()
        n = float(v.numel())
        mean = s[0] / n
        return mean, float(np.sqrt(max(s[1] / n - mean * mean, 0.0)))

    @staticmethod
    def _upload(vol_zyx, device):
        a = np.ascontiguousarray(vol_zyx)
        if np.dtype(a.dtype) not in _DEVICE_DT:
            a = a.astype(np.float32)
        if not a.flags.writeable:       # np.frombuffer views of the file image are read-only
            a = a.copy()
        return torch.from_numpy(a).to(device)

    @classmethod
    def from_volume(cls, vol_zyx, device, image_size, modality="MR", ct_mean=None, ct_std=None, tile_z_dim=3,
                    labels_zyx=None, info=None, scaling=(1.0, 0.0)):
        """vol_zyx: numpy [z, y, x] as read from the file (any of int16 / float32 / uint8 / int32; others are cast)."""
        if modality not in ("MR", "CT"):
            raise ValueError(f"modality must be 'MR' or 'CT', got {modality}")   # get_normalize_op, dataset_utils.py:110-127
        v = cls._upload(vol_zyx, device)
        dt = _TORCH_DT[v.dtype]
        Z, H, W = v.shape
        if modality == "MR":
            s = ops.volume_stats(v, dt, scaling[0], scaling[1]).cpu().numpy()
            mean = s[0] / v.numel()
            std = float(np.sqrt(max(s[1] / v.numel() - mean * mean, 0.0)))
        else:
            if ct_mean is None or ct_std is None:
                raise ValueError("CT normalisation needs the fold's global ct_mean / ct_std (get_CT_statistics)")
            mean, std = float(ct_mean), float(ct_std)
        imgs = ops.volume_slices(v, dt, Z, H, W, scaling[0], scaling[1], mean, 1.0 / std, image_size, tile_z_dim, 0)
        labs = None
        if labels_zyx is not None:
            lv = cls._upload(labels_zyx, device)
            ldt = _TORCH_DT[lv.dtype]
            assert tuple(lv.shape) == (Z, H, W), "image and label volumes differ in shape"
            labs = ops.volume_slices(lv, ldt, Z, H, W, 1.0, 0.0, 0.0, 1.0, image_size, 1, 1)[:, 0]
        return cls(imgs, labs, mean, std, info)

    @classmethod
    def from_nifti(cls, img_path, device, image_size, label_path=None, **kw):
        vol, info = read_nifti(img_path, peel_info=True)
        scaling = (1.0, 0.0)          # read_nifti already applied a non-trivial slope / intercept
        lab = read_nifti(label_path) if label_path is not None else None
        return cls.from_volume(vol, device, image_size, labels_zyx=lab, info=info, scaling=scaling, **kw)
