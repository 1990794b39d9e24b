"""CPU: host-side logic that needs no kernel launch -- packed-weight caches are dropped by a PARENT's load_state_dict,
the two-file NIfTI form, the metric export."""
import gzip

import numpy as np
import pytest
import torch
import torch.nn as nn


def test_parent_load_state_dict_drops_child_caches():
    """nn.Module.load_state_dict recurses through _load_from_state_dict and never calls a child's load_state_dict: the
    caches must be invalidated in that hook (Sam.load_state_dict / FewShotSeg.load_state_dict after a forward)."""
    from protosam_amd.segment_anything import sam_model_registry
    sam = sam_model_registry["vit_b"](encoder_depth=1)
    sam.image_encoder._packed = "stale"
    sam.mask_decoder._cache = "stale"
    sam.prompt_encoder._cache = "stale"
    sam.load_state_dict(sam.state_dict())                        # parent load
    assert sam.image_encoder._packed is None and sam.mask_decoder._cache is None and sam.prompt_encoder._cache is None

    holder = nn.Module()
    holder.sam = sam
    sam.image_encoder._packed = "stale"
    holder.load_state_dict(holder.state_dict())                  # grand-parent load
    assert sam.image_encoder._packed is None


def test_fewshot_load_state_dict_drops_support_cache_and_encoder_pack():
    from protosam_amd.grid_proto_fewshot import FewShotSeg
    cfg = {"which_model": "dinov2_b14", "cls_name": "grid_proto", "proto_grid_size": 8, "lora": 0, "align": False,
           "debug": False, "use_coco_init": False, "encoder_depth": 1}
    m = FewShotSeg(252, None, cfg)
    m._sup_cache = ["stale"]
    m.encoder._packed = "stale"
    e0 = m.encoder._weights_epoch if hasattr(m.encoder, "_weights_epoch") else 0
    m.load_state_dict(m.state_dict())
    assert m._sup_cache == [] and m.encoder._packed is None
    # loading the encoder alone must invalidate the banks built from it as well: its epoch is part of the cache key
    m.encoder.load_state_dict(m.encoder.state_dict())
    assert m.encoder._weights_epoch > e0 + 1


def test_two_file_nifti(tmp_path):
    from oracle.slice_io import nifti_bytes
    from protosam_amd.slice_io import read_nifti
    vol = (np.random.RandomState(3).randn(4, 6, 8) * 100).astype(np.int16)
    raw = bytearray(nifti_bytes(vol))
    off = int(np.frombuffer(bytes(raw[108:112]), dtype="<f4")[0])
    hdr = bytearray(raw[:off])
    hdr[344:348] = b"ni1\0"
    hdr[108:112] = np.float32(0).tobytes()                       # vox_offset of the .img
    (tmp_path / "v.hdr").write_bytes(bytes(hdr))
    with pytest.raises(ValueError):                              # no .img beside it: never reinterpret header bytes as voxels
        read_nifti(str(tmp_path / "v.hdr"))
    (tmp_path / "v.img").write_bytes(bytes(raw[off:]))
    assert np.array_equal(read_nifti(str(tmp_path / "v.hdr")), vol)
    (tmp_path / "v.img").write_bytes(bytes(raw[off:-10]))
    with pytest.raises(ValueError):                              # truncated voxel data
        read_nifti(str(tmp_path / "v.hdr"))
    trunc = tmp_path / "t.nii.gz"
    gzip.open(trunc, "wb").write(bytes(raw[:-4]))
    with pytest.raises(ValueError):
        read_nifti(str(trunc))


def test_module_level_caches_follow_loads():
    """The blocks / two-way transformer parts pack their own weights (their stand-alone `forward`s use the packs too): a parent's
    load_state_dict and `.to()` must drop every one of them."""
    from protosam_amd.segment_anything import sam_model_registry
    sam = sam_model_registry["vit_b"](encoder_depth=1)
    blk = sam.image_encoder.blocks[0]
    tb = sam.mask_decoder.transformer.layers[0]
    blk._pk, tb._cache, tb.self_attn._cache = {"grid": 64}, "stale", {(): "stale"}
    sam.load_state_dict(sam.state_dict())
    assert blk._pk is None and tb._cache is None and tb.self_attn._cache == {}
    blk._pk, tb._cache, tb._ws, sam.mask_decoder.transformer._ws = {"grid": 64}, "stale", {1: 2}, {1: 2}
    sam.float()                                                  # any _apply (.to / .cuda / .half)
    assert blk._pk is None and tb._cache is None and tb._ws == {} and sam.mask_decoder.transformer._ws == {}


def test_split_weight_scale_stays_inside_fp16():
    """ops.split_weight_f16 refuses a scale that would push a weight past the fp16 range (hi = inf, lo = -inf, NaN products in
    psam_gemm_f32x3); ops.split_scale_for picks the largest power of two <= 2^8 that fits."""
    from protosam_amd import ops
    w = torch.tensor([[0.05, -0.3], [300.0, 1e-4]])
    with pytest.raises(ValueError):
        ops.split_weight_f16(w, 256.0)
    s = ops.split_scale_for(w, 256.0)
    assert s == 128.0
    hi, lo = ops.split_weight_f16(w, s)
    assert torch.isfinite(hi.float()).all() and torch.isfinite(lo.float()).all()
    back = (hi.double() + lo.double()) / s
    assert (back - w.double()).abs().max() <= w.abs().max().item() * 2.0 ** -20
    assert ops.split_scale_for(torch.tensor([0.1, -0.2]), 256.0) == 256.0
    with pytest.raises(ValueError):
        ops.split_weight_f16(torch.tensor([float("inf")]))


def test_t2i_split_mode_one_means_one_range(monkeypatch):
    from protosam_amd import ops
    monkeypatch.setenv("PSAM_T2I_SPLIT", "1")
    assert ops.t2i_split(1, 8, 7, 4096) == 1
    monkeypatch.setenv("PSAM_T2I_SPLIT", "4")
    assert ops.t2i_split(1, 8, 7, 4096) == 4
    monkeypatch.setenv("PSAM_T2I_SPLIT", "auto")
    assert ops.t2i_split(1, 8, 7, 4096) == 16 and ops.t2i_split(27, 8, 7, 4096) == 1
