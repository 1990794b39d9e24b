"""CPU: automatic-mask-generator host logic (point grids, RLE, NMS, the `custom_points` label rule) and the oracle's
generator against the REFERENCE's recorded records (tests/golden/reference_outputs.npz, `amg_*` arrays written by
oracle/validate_against_reference.py from the vendored SamAutomaticMaskGenerator / models/SamWrapper.py)."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_outputs.npz")


def test_point_grid_and_batches():
    from oracle import amg as oamg
    from protosam_amd.segment_anything.utils import amg
    for n in (1, 3, 8, 32):
        np.testing.assert_array_equal(amg.build_point_grid(n), oamg.point_grid(n))
    g = amg.build_point_grid(32) * 1024
    assert g.shape == (1024, 2) and g[0].tolist() == [16.0, 16.0] and g[1].tolist() == [48.0, 16.0]
    grids = amg.build_all_layer_point_grids(32, 2, 2)
    assert [len(x) for x in grids] == [1024, 256, 64]
    chunks = list(amg.batch_iterator(64, g, np.arange(1024)))
    assert len(chunks) == 16 and chunks[-1][0].shape == (64, 2)
    assert [len(c[0]) for c in amg.batch_iterator(5, np.arange(12))] == [5, 5, 2]
    with pytest.raises(AssertionError):
        list(amg.batch_iterator(4, np.arange(3), np.arange(4)))


def test_rle_roundtrip_and_area():
    from oracle import amg as oamg
    from protosam_amd.segment_anything.utils import amg
    rng = np.random.RandomState(0)
    cases = [rng.rand(17, 23) > 0.5, np.zeros((5, 7), bool), np.ones((4, 3), bool), rng.rand(1, 9) > 0.3,
             rng.rand(9, 1) > 0.3]
    first_set = np.zeros((6, 6), bool)
    first_set[0, 0] = True
    cases.append(first_set)
    for m in cases:
        r = amg.mask_to_rle(m)
        assert r == oamg.rle(m)
        assert r["size"] == list(m.shape) and sum(r["counts"]) == m.size
        np.testing.assert_array_equal(amg.rle_to_mask(r), m)
        np.testing.assert_array_equal(oamg.rle_to_mask(r), m)
        assert amg.area_from_rle(r) == int(m.sum())
    assert amg.mask_to_rle(first_set)["counts"][0] == 0          # a set first pixel starts with an empty zero-run
    assert amg.box_xyxy_to_xywh(np.array([3, 4, 10, 20])).tolist() == [3, 4, 7, 16]


def test_nms_known_answers():
    from protosam_amd.segment_anything.utils.amg import nms_xyxy
    boxes = np.array([[0, 0, 10, 10], [1, 1, 11, 11], [20, 20, 30, 30], [0, 0, 10, 10], [5, 5, 6, 6]], np.float32)
    scores = np.array([0.9, 0.8, 0.7, 0.95, 0.1], np.float32)
    # box 3 (best) suppresses 0 (IoU 1) and 1 (IoU 81/119 = 0.68 <= 0.7 -> kept)
    assert nms_xyxy(boxes, scores, 0.7).tolist() == [3, 1, 2, 4]
    assert nms_xyxy(boxes, scores, 0.6).tolist() == [3, 2, 4]
    assert nms_xyxy(boxes, scores, 1.0).tolist() == [3, 0, 1, 2, 4]            # IoU > 1 never happens: a pure sort
    assert nms_xyxy(np.zeros((0, 4)), np.zeros(0), 0.7).tolist() == []
    # degenerate (zero-area) boxes: 0/0 = nan is not > thr, nothing is suppressed; ties keep the lower index first
    z = np.zeros((3, 4), np.float32)
    assert nms_xyxy(z, np.array([0.5, 0.5, 0.5], np.float32), 0.7).tolist() == [0, 1, 2]


def test_nms_matches_oracle_restatement():
    from oracle import amg as oamg
    from protosam_amd.segment_anything.utils.amg import nms_xyxy
    rng = np.random.RandomState(3)
    for trial in range(6):
        n = 60
        xy = rng.randint(0, 200, (n, 2))
        wh = rng.randint(0, 120, (n, 2))
        boxes = np.concatenate([xy, xy + wh], 1).astype(np.int64)
        boxes[rng.rand(n) < 0.2] = boxes[0]                                     # exact duplicates
        scores = rng.rand(n).astype(np.float32)
        scores[::7] = scores[0]                                                 # score ties
        for thr in (0.3, 0.7, 0.95):
            a = nms_xyxy(boxes, scores, thr)
            b = oamg.batched_nms(torch.as_tensor(boxes).float(), torch.as_tensor(scores),
                                 torch.zeros(n, dtype=torch.long), thr)
            assert a.tolist() == b.tolist()


def test_custom_points_label_rule():
    """automatic_mask_generator.py:52,280: the default is the truthy string "false" -> the second half of every batch
    is labelled negative; an odd batch cannot be labelled."""
    from protosam_amd.segment_anything import SamAutomaticMaskGenerator, sam_model_registry
    sam = sam_model_registry["vit_b"](encoder_depth=1)
    g = SamAutomaticMaskGenerator(sam, points_per_side=4, points_per_batch=8)
    assert g._point_labels(16).tolist() == [1, 1, 1, 1, 0, 0, 0, 0] * 2
    assert g._point_labels(12).tolist() == [1, 1, 1, 1, 0, 0, 0, 0, 1, 1, 0, 0]
    with pytest.raises(ValueError):
        g._point_labels(11)
    g = SamAutomaticMaskGenerator(sam, points_per_side=4, points_per_batch=8, custom_points=False)
    assert g._point_labels(11).tolist() == [1] * 11
    g = SamAutomaticMaskGenerator(sam, points_per_side=8, crop_n_layers=2, crop_n_points_downscale_factor=2,
                                  min_mask_region_area=10)
    assert [len(p) for p in g.point_grids] == [64, 16, 4]                       # one grid per crop layer (:119-124)
    assert not g._fast_path(np.zeros((1024, 1024, 3), np.uint8))
    g = SamAutomaticMaskGenerator(sam, points_per_side=8)
    assert g._fast_path(np.zeros((1024, 768, 3), np.uint8)) and not g._fast_path(np.zeros((48, 44, 3), np.uint8))
    with pytest.raises(ImportError):                                            # pycocotools is absent, as it may be for the
        SamAutomaticMaskGenerator(sam, points_per_side=8, output_mode="coco_rle")   # reference (:116-117): same error type
    with pytest.raises(AssertionError):
        SamAutomaticMaskGenerator(sam, points_per_side=None)
    with pytest.raises(AssertionError):
        SamAutomaticMaskGenerator(sam, output_mode="png")


def test_oracle_generator_reproduces_reference_records():
    from oracle import amg as oamg
    from protosam_amd import synth_cases as gi
    from protosam_amd.segment_anything import sam_model_registry
    from protosam_amd.synth import synth_state_dict
    gold = np.load(GOLD)
    sd = synth_state_dict(sam_model_registry["vit_b"](encoder_depth=gi.AMG_ENCODER_DEPTH), gi.AMG_SEED)
    img, label = gi.amg_case()
    t_iou, t_stab = gold["amg_thresholds"]
    best, bi, ious, anns = oamg.sam_wrapper_forward(img, label, sd, encoder_depth=gi.AMG_ENCODER_DEPTH,
                                                    pred_iou_thresh=float(t_iou), stability_score_thresh=float(t_stab),
                                                    **gi.AMG_ARGS)
    assert len(anns) == len(gold["amg_pred_iou"])
    np.testing.assert_allclose([a["predicted_iou"] for a in anns], gold["amg_pred_iou"], atol=2e-5)
    np.testing.assert_allclose([a["stability_score"] for a in anns], gold["amg_stability"], atol=2e-5)
    np.testing.assert_array_equal(np.array([a["bbox"] for a in anns]), gold["amg_bbox"])
    assert np.abs(np.array([a["area"] for a in anns]) - gold["amg_area"]).max() <= 4
    np.testing.assert_array_equal(np.array([a["point_coords"][0] for a in anns]), gold["amg_points"])
    assert bi == int(gold["amg_best_index"][0])
    ref_best = np.unpackbits(gold["amg_best_mask_bits"]).reshape(1024, 1024).astype(bool)
    assert int((best != ref_best).sum()) <= 4


@pytest.mark.parametrize("tag,extra", [("all", dict(crop_nms_thresh=1.0)), ("nms", dict())])
def test_oracle_crops_and_small_regions_reproduce_reference_records(tag, extra):
    """crop_n_layers = 1 + min_mask_region_area (automatic_mask_generator.py:194-380) on the 48 x 44 image: the oracle against
    the records the REFERENCE's generator produced (oracle/validate_against_reference.py check_amg)."""
    from oracle import amg as oamg
    from protosam_amd import synth_cases as gi
    from protosam_amd.segment_anything import sam_model_registry
    from protosam_amd.synth import synth_state_dict
    gold = np.load(GOLD)
    sd = synth_state_dict(sam_model_registry["vit_b"](encoder_depth=gi.AMG_ENCODER_DEPTH), gi.AMG_SEED)
    img = gi.amg_small_case()
    H, W = img.shape[:2]
    anns = oamg.generate(img, sd, encoder_depth=gi.AMG_ENCODER_DEPTH, **dict(gi.AMG_CROP_ARGS, **extra))
    n = len(gold[f"amgc_{tag}_pred_iou"])
    assert len(anns) == n
    np.testing.assert_allclose([a["predicted_iou"] for a in anns], gold[f"amgc_{tag}_pred_iou"], atol=2e-5)
    np.testing.assert_allclose([a["stability_score"] for a in anns], gold[f"amgc_{tag}_stability"], atol=2e-5)
    np.testing.assert_array_equal(np.array([a["crop_box"] for a in anns]), gold[f"amgc_{tag}_crop_box"])
    np.testing.assert_array_equal(np.array([a["point_coords"][0] for a in anns]), gold[f"amgc_{tag}_points"])
    ref_masks = np.unpackbits(gold[f"amgc_{tag}_mask_bits"], axis=1)[:, :H * W].reshape(n, H, W).astype(bool)
    diff = [int((a["segmentation"] != m).sum()) for a, m in zip(anns, ref_masks)]
    assert max(diff) <= 2 and sum(d == 0 for d in diff) >= n - 2           # (thread-count dependent fp32 summation order)
    if max(diff) == 0:
        np.testing.assert_array_equal(np.array([a["bbox"] for a in anns]), gold[f"amgc_{tag}_bbox"])
        np.testing.assert_array_equal(np.array([a["area"] for a in anns]), gold[f"amgc_{tag}_area"])


def test_crop_boxes_and_edge_rule():
    """Host helpers of the crop path against the oracle's restatements and known answers (utils/amg.py:78-88, :202-237)."""
    from oracle import amg as oamg
    from protosam_amd.segment_anything.utils.amg import generate_crop_boxes, is_box_near_crop_edge
    for size in ((48, 44), (1024, 1024), (480, 640), (333, 1000)):
        for n_layers in (0, 1, 2):
            b, l = generate_crop_boxes(size, n_layers, 512 / 1500)
            bo, lo = oamg.crop_boxes_for(size, n_layers, 512 / 1500)
            assert b == [list(x) for x in bo] and l == list(lo)
            assert len(b) == sum(4 ** i for i in range(n_layers + 1)) and b[0] == [0, 0, size[1], size[0]]
    b, _ = generate_crop_boxes((48, 44), 1, 512 / 1500)
    assert b[1:] == [[0, 0, 30, 32], [0, 16, 30, 48], [14, 0, 44, 32], [14, 16, 44, 48]] or len(b) == 5
    # a box touching the crop's right edge, far from the image's right edge -> dropped; near both -> kept
    crop, orig = [100, 100, 400, 400], [0, 0, 1000, 1000]
    boxes = np.array([[50, 50, 299, 120], [50, 50, 120, 120], [0, 50, 120, 120]])
    assert is_box_near_crop_edge(boxes, crop, orig).tolist() == [True, False, True]
    assert is_box_near_crop_edge(boxes, [0, 0, 300, 300], [0, 0, 310, 310]).tolist() == [False, False, False]
    rng = np.random.default_rng(0)
    bx = rng.integers(0, 300, (200, 4))
    np.testing.assert_array_equal(is_box_near_crop_edge(bx, crop, orig),
                                  oamg.box_near_crop_edge(torch.as_tensor(bx), crop, orig).numpy())
