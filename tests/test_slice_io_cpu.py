"""CPU: NIfTI reader / writer of protosam_amd/slice_io.py against independently built files, and the oracle's cv2.resize
restatement against torch's interpolation."""
import gzip
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F


def test_nifti_reader_on_independent_bytes(tmp_path):
    from oracle.slice_io import nifti_bytes
    from protosam_amd.slice_io import read_nifti
    rng = np.random.RandomState(1)
    vol = (rng.randn(6, 10, 12) * 300).astype(np.int16)
    for endian, gz in (("<", False), (">", False), ("<", True)):
        p = tmp_path / ("v.nii.gz" if gz else f"v{ord(endian)}.nii")
        data = nifti_bytes(vol, spacing=(0.7, 0.8, 2.5), qoffset=(10.0, -20.0, 30.0), endian=endian)
        (gzip.open if gz else open)(p, "wb").write(data)
        arr, info = read_nifti(str(p), peel_info=True)
        assert arr.dtype == np.int16 and np.array_equal(arr, vol) and info["array_size"] == (6, 10, 12)
        np.testing.assert_allclose(info["spacing"], (0.7, 0.8, 2.5), rtol=1e-6)
        # ITK reports LPS: NIfTI's (x, y) are flipped
        np.testing.assert_allclose(info["origin"], (-10.0, 20.0, 30.0), rtol=1e-6)
        np.testing.assert_allclose(info["direction"], (-1, 0, 0, 0, -1, 0, 0, 0, 1), atol=1e-7)
        assert np.array_equal(read_nifti(str(p)), vol)
    # scl_slope / scl_inter rescale the voxels to float32 (as ITK's reader does)
    p = tmp_path / "s.nii"
    open(p, "wb").write(nifti_bytes(vol, slope=0.5, inter=-3.0))
    arr = read_nifti(str(p))
    assert arr.dtype == np.float32 and np.array_equal(arr, vol.astype(np.float32) * np.float32(0.5) + np.float32(-3.0))
    # a 90-degree rotation about z as a quaternion
    p = tmp_path / "q.nii"
    open(p, "wb").write(nifti_bytes(vol, quatern=(0.0, 0.0, np.sqrt(0.5))))
    _, info = read_nifti(str(p), peel_info=True)
    np.testing.assert_allclose(np.array(info["direction"]).reshape(3, 3), [[0, 1, 0], [-1, 0, 0], [0, 0, 1]], atol=1e-6)
    # malformed files
    open(tmp_path / "bad.nii", "wb").write(b"\0" * 400)
    with pytest.raises(ValueError):
        read_nifti(str(tmp_path / "bad.nii"))
    open(tmp_path / "short.nii", "wb").write(b"\0" * 10)
    with pytest.raises(ValueError):
        read_nifti(str(tmp_path / "short.nii"))


def test_nifti_writer_roundtrip(tmp_path):
    from protosam_amd.slice_io import read_nifti, write_nifti
    rng = np.random.RandomState(2)
    info = {"spacing": (0.8, 0.9, 3.0), "origin": (-100.0, 50.0, 12.5),
            "direction": (0.0, 1.0, 0.0, -1.0, 0.0, 0.0, 0.0, 0.0, 1.0)}
    for arr in ((rng.rand(4, 8, 8) > 0.5), (rng.randn(3, 5, 7) * 50).astype(np.int16), rng.randn(2, 4, 6).astype(np.float32),
                rng.randn(2, 4, 6)):
        for name in ("m.nii", "m.nii.gz"):
            write_nifti(str(tmp_path / name), arr, info)
            back, got = read_nifti(str(tmp_path / name), peel_info=True)
            exp = arr.astype(np.uint8) if arr.dtype == bool else (arr.astype(np.float32) if arr.dtype == np.float64 else arr)
            assert back.dtype == exp.dtype and np.array_equal(back, exp)
            np.testing.assert_allclose(got["spacing"], info["spacing"], rtol=1e-6)
            np.testing.assert_allclose(got["origin"], info["origin"], rtol=1e-6)
            np.testing.assert_allclose(got["direction"], info["direction"], atol=1e-6)
    write_nifti(str(tmp_path / "plain.nii"), np.zeros((2, 3, 4), np.uint8))
    _, got = read_nifti(str(tmp_path / "plain.nii"), peel_info=True)
    assert got["spacing"] == (1.0, 1.0, 1.0) and got["direction"] == (1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0)
    with pytest.raises(ValueError):
        write_nifti(str(tmp_path / "x.nii"), np.zeros((3, 4)))


def test_resize_restatements_vs_torch():
    """The oracle's cv2.INTER_LINEAR / INTER_NEAREST restatements equal torch's bilinear (align_corners=False, no
    antialias) / nearest interpolation, up- and down-scaling, to fp32 rounding."""
    from oracle.slice_io import prepare_scan, resize_linear, resize_nearest
    rng = np.random.RandomState(3)
    for (H, W, S) in ((37, 53, 64), (300, 260, 128), (64, 64, 64), (50, 70, 33)):
        img = rng.randn(H, W).astype(np.float32)
        ref = F.interpolate(torch.from_numpy(img)[None, None], size=(S, S), mode="bilinear", align_corners=False)[0, 0]
        # torch derives the source coordinate from a float32 scale, cv2 from a double one: ~1e-6 in the weights
        assert np.abs(resize_linear(img, S) - ref.numpy()).max() < 5e-5
        lab = rng.randint(0, 5, (H, W)).astype(np.float32)
        refn = F.interpolate(torch.from_numpy(lab)[None, None], size=(S, S), mode="nearest")[0, 0]
        assert np.array_equal(resize_nearest(lab, S), refn.numpy())
    vol = (rng.randn(5, 40, 48) * 200).astype(np.int16)
    imgs, labs = prepare_scan(vol, 32, "MR", labels_zyx=(vol > 100).astype(np.uint8))
    assert imgs.shape == (5, 3, 32, 32) and labs.shape == (5, 32, 32) and imgs.dtype == np.float32
    assert np.array_equal(imgs[:, 0], imgs[:, 2]) and set(np.unique(labs)) <= {0.0, 1.0}
