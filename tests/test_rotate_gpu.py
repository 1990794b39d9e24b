"""Rotation test-time augmentation (ProtoSAM.forward(..., degrees_rotate != 0)): the two HIP kernels against the restated
torchvision tensor ops (oracle/rotate.py; parity unpinned - torchvision is absent), the two helpers against util/utils.py's
restatement, and the pipeline end to end."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _rand(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g)


@pytest.mark.parametrize("angle", [15, -30, 45, 90, 180, 7.5])
@pytest.mark.parametrize("hw", [(64, 64), (96, 130), (512, 512)])
def test_rotate_tensor_no_crop_vs_oracle(dev, angle, hw):
    """rotate(expand=True, NEAREST) + antialiased resize back to H x W; index work is exact up to fp rounding ties of the
    sampling coordinate (a handful of pixels at most), the resize within 1e-5."""
    from oracle import rotate as orot
    from protosam_amd import ops, rotate as prot
    h, w = hw
    x = _rand((2, 3, h, w), 5)
    ref_rot = orot.tv_rotate(x, angle, expand=True)
    matrix = prot._inverse_rotation_matrix(-angle)
    assert prot._affine_output_size(matrix, w, h) == orot.affine_output_size(matrix, w, h)
    got_rot = prot._rotate(x.to(dev), angle, expand=True).cpu()
    assert got_rot.shape == ref_rot.shape
    bad = (got_rot != ref_rot).float().mean().item()
    assert bad < 2e-4, f"{bad:.2e} of the rotated pixels differ"
    ref, (rh, rw) = orot.rotate_tensor_no_crop(x, angle)
    got, (gh, gw) = prot.rotate_tensor_no_crop(x.to(dev), angle)
    assert (rh, rw) == (gh, gw) and got.shape == ref.shape
    # where a nearest-sample tie flipped a source pixel the resized value moves; compare through the same rotated input
    torch.testing.assert_close(ops.resize_aa(ref_rot.to(dev).contiguous(), h, w).cpu(), ref, rtol=1e-5, atol=2e-5)
    assert (got.cpu() - ref).abs().mean().item() < 1e-4


@pytest.mark.parametrize("sizes", [((50, 70), (20, 31)), ((40, 40), (100, 90)), ((512, 512), (724, 724)), ((33, 47), (33, 47))])
def test_resize_aa_vs_torch(dev, sizes):
    """anti-aliased bilinear, down- and up-scaling, against aten's _upsample_bilinear2d_aa (CPU)."""
    from protosam_amd import ops
    (h, w), (oh, ow) = sizes
    x = _rand((1, 2, h, w), 9)
    ref = torch.nn.functional.interpolate(x, size=[oh, ow], mode="bilinear", align_corners=False, antialias=True)
    got = ops.resize_aa(x.to(dev), oh, ow).cpu()
    torch.testing.assert_close(got, ref, rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("angle", [15, -30, 90])
def test_reverse_tensor_vs_oracle(dev, angle):
    from oracle import rotate as orot
    from protosam_amd import rotate as prot
    x = _rand((1, 3, 128, 128), 3)
    _, (rh, rw) = orot.rotate_tensor_no_crop(x, angle)
    logits = _rand((1, 2, 128, 128), 4)
    ref = orot.reverse_tensor(logits, rh, rw, -angle)
    got = prot.reverse_tensor(logits.to(dev), rh, rw, -angle).cpu()
    assert got.shape == ref.shape
    # same resized tensor up to 1e-5, then a nearest gather: values agree except at coordinate ties
    close = ((got - ref).abs() < 1e-4).float().mean().item()
    assert close > 0.9995, close
    # identity at 0 degrees (the reference caller's only value)
    y, sz = prot.rotate_tensor_no_crop(x.to(dev), 0)
    assert sz == (128, 128) and torch.equal(y.cpu(), x)


def test_protosam_forward_with_rotation_vs_oracle(dev):
    """ProtoSAM.forward(q, inp, degrees_rotate=15): coarse model on the rotated query, logits rotated back, then the
    usual prompts + SAM on the ORIGINAL image (ProtoSAM.py:544-556)."""
    from oracle import alp as oalp, dinov2 as odino, glue, rotate as orot
    from protosam_amd.protosam import InputFactory, TYPE_ALPNET
    from protosam_amd.synth import synth_pair, synth_state_dict
    from tests.test_protosam_gpu import _build, _dice
    sam_depth, dino_depth, deg = 3, 12, 15
    kw = dict(use_bbox=True, use_points=True, point_mode="both", use_cca=False)
    model, alp_sd = _build(dev, f"random:vit_b:1234:{sam_depth}", dino_depth, **kw)
    sam_sd = {k: v.cpu() for k, v in synth_state_dict(model.sam, 1234).items()}
    s_img, s_m, q_img, _ = synth_pair(512, seed=0)
    inp = InputFactory.create_input(TYPE_ALPNET, q_img, support_images=[s_img], support_labels=[s_m], isval=True,
                                    val_wsize=2)
    inp.to(dev)
    pred, scores = model(q_img.to(dev), inp, degrees_rotate=deg)
    pred0, _ = model(q_img.to(dev), inp)
    enc_sd = {k[len("encoder."):]: v for k, v in alp_sd.items() if k.startswith("encoder.")}
    enc = lambda im: odino.forward_features(im, enc_sd, "dinov2_b14", depth=dino_depth)["x_norm_patchtokens"]  # noqa
    rq, (rh, rw) = orot.rotate_tensor_no_crop(q_img, deg)
    logits_rot = oalp.fewshot_forward(enc, s_img, s_m, rq, 512)
    logits_ref = orot.reverse_tensor(logits_rot, rh, rw, -deg)
    pred_ref, scores_ref = glue.protosam_forward(q_img, logits_ref, sam_sd, "vit_b", use_bbox=True, use_points=True,
                                                 point_mode="both", use_cca=False, encoder_depth=sam_depth)
    d = _dice(pred.cpu(), pred_ref)
    print(f"rotation {deg} deg: Dice vs oracle {d:.5f}, Dice vs unrotated {_dice(pred.cpu(), pred0.cpu()):.4f}, "
          f"scores {np.abs(np.array(scores) - np.array(scores_ref)).max():.2e}")
    assert pred.shape == pred_ref.shape and d > 0.99
