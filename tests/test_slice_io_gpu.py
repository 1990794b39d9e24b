"""GPU: scan volume -> normalised / resized / tiled slices on the device (psam_volume_stats, psam_volume_slices) against the
oracle's restatement of the reference's host chain, from raw arrays and from NIfTI files, and straight into the runner."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype,modality,shape,S", [(np.int16, "MR", (9, 256, 256), 512), (np.float32, "MR", (5, 300, 260), 256),
                                                    (np.int16, "CT", (7, 512, 512), 512), (np.uint8, "MR", (4, 97, 131), 64),
                                                    (np.int32, "CT", (3, 640, 640), 320)])
def test_scan_slices_vs_oracle(dev, dtype, modality, shape, S):
    from oracle.slice_io import prepare_scan
    from protosam_amd.slice_io import ScanSlices
    rng = np.random.RandomState(shape[0])
    base = rng.randn(*shape) * 300 + 100
    vol = np.clip(base, 0, 255).astype(dtype) if dtype == np.uint8 else base.astype(dtype)
    lab = (rng.rand(*shape) > 0.7).astype(np.uint8) * rng.randint(1, 4, shape).astype(np.uint8)
    kw = dict(ct_mean=87.5, ct_std=311.0) if modality == "CT" else {}
    ss = ScanSlices.from_volume(vol, dev, S, modality=modality, labels_zyx=lab, **kw)
    ref_i, ref_l = prepare_scan(vol, S, modality, labels_zyx=lab, **kw)
    assert ss.images.shape == (shape[0], 3, S, S) and ss.images.dtype == torch.float32 and ss.images.is_cuda
    if modality == "MR":
        v32 = np.float32(vol)
        assert abs(ss.mean - float(v32.mean(dtype=np.float64))) < 1e-6 * max(1.0, abs(v32.mean()))
        assert abs(ss.std - float(v32.std(dtype=np.float64))) < 1e-6 * v32.std()
    err = np.abs(ss.images.cpu().numpy() - ref_i).max()
    print(f"{np.dtype(dtype).name} {modality} {shape} -> {S}: max abs err {err:.2e} on values up to {np.abs(ref_i).max():.2f}")
    assert err < 2e-5 * max(1.0, np.abs(ref_i).max())
    assert np.array_equal(ss.labels.cpu().numpy(), ref_l)
    m, s = ScanSlices.volume_stats(vol, dev)
    assert abs(m - float(np.float64(vol).mean())) < 1e-9 * max(1.0, abs(m)) and abs(s - float(np.float64(vol).std())) < 1e-7 * s


def test_nifti_to_masks_roundtrip(dev, tmp_path):
    """A NIfTI scan streams through ScanSlices into the slice runner; the predicted mask volume is written back with the
    scan's geometry and read again."""
    from oracle.slice_io import nifti_bytes, prepare_scan
    from protosam_amd.runner import build_protosam, run_slices, support_set
    from protosam_amd.slice_io import ScanSlices, read_nifti, write_nifti
    from protosam_amd.synth import synth_volume
    vol, lab = synth_volume(12, 256, seed=4, kind="mri")
    raw = (vol.numpy() * 400 + 500).astype(np.int16)
    open(tmp_path / "scan.nii", "wb").write(nifti_bytes(raw, spacing=(1.2, 1.2, 5.0), qoffset=(3.0, 4.0, 5.0)))
    open(tmp_path / "lab.nii", "wb").write(nifti_bytes(lab.numpy().astype(np.uint8)))
    ss = ScanSlices.from_nifti(str(tmp_path / "scan.nii"), dev, 512, label_path=str(tmp_path / "lab.nii"), modality="MR")
    ref_i, ref_l = prepare_scan(raw, 512, "MR", labels_zyx=lab.numpy().astype(np.uint8))
    assert np.abs(ss.images.cpu().numpy() - ref_i).max() < 1e-4 and np.array_equal(ss.labels.cpu().numpy(), ref_l)
    model, _ = build_protosam(dev, sam_type="vit_b", image_size=512, seed=1234, sam_depth=1)
    sup_imgs, sup_masks = support_set(ss.images[:, 0], ss.labels)
    zs = [4, 5, 6, 7]
    masks, _ = run_slices(model, ss.images[:, 0], sup_imgs, sup_masks, zs, dev, batch=4)
    assert masks.shape == (4, 512, 512) and masks.dtype == torch.uint8
    write_nifti(str(tmp_path / "pred.nii.gz"), masks.cpu().numpy(), ss.info)
    back, info = read_nifti(str(tmp_path / "pred.nii.gz"), peel_info=True)
    assert np.array_equal(back, masks.cpu().numpy())
    np.testing.assert_allclose(info["spacing"], (1.2, 1.2, 5.0), rtol=1e-6)
    np.testing.assert_allclose(info["origin"], ss.info["origin"], rtol=1e-6)
