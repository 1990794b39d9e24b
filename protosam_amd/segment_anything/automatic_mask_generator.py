"""`SamAutomaticMaskGenerator` (models/segment_anything/automatic_mask_generator.py:35-373) on the HIP path.

Same constructor arguments, record keys and filtering rules as the reference's generator; what differs is where the work
happens:
  * all grid points of a crop go through the two-way decoder in chunks of `decode_chunk` prompts (the reference: 64 per
    call, :255-259); the decoder is per-prompt independent, so chunking does not change results;
  * the reference up-samples every candidate to full resolution ([64*3, H, W] fp32) to take three counts and a box from
    it (:293-310). `psam_mask_stats` computes them straight from the 256x256 logits; nothing full-size is written;
  * box NMS (:262-268) runs on the host over the few hundred survivors of the score filters;
  * only the survivors of NMS are binarised (`psam_mask_binarize`), and they stay on the device until a caller asks
    for numpy (`generate`) - `SamWrapper` scores them against the label on the device and downloads one mask.
The intermediate uncompressed RLE (:312-313) only exists when `output_mode` asks for it.

`custom_points` keeps the reference's default, the *string* "false" (:52), and the reference's truthiness test (:280):
with the default, the second half of every `points_per_batch` batch is labelled as NEGATIVE points. `custom_points=False`
gives upstream SAM's behaviour.

Not on this path (raise NotImplementedError): crop_n_layers > 0 (needs PIL crops + re-encode per crop),
min_mask_region_area > 0 (cv2 in the reference, utils/amg.py:267-291) and output_mode "coco_rle" (pycocotools).
"""
import numpy as np
import torch

from .. import ops
from .predictor import SamPredictor
from .utils.amg import (area_from_rle, box_xyxy_to_xywh, build_all_layer_point_grids, mask_to_rle, nms_xyxy)


class SamAutomaticMaskGenerator:
    def __init__(self, model, points_per_side=32, points_per_batch=64, pred_iou_thresh=0.88,
                 stability_score_thresh=0.95, stability_score_offset=1.0, box_nms_thresh=0.7, crop_n_layers=0,
                 crop_nms_thresh=0.7, crop_overlap_ratio=512 / 1500, crop_n_points_downscale_factor=1,
                 point_grids=None, min_mask_region_area=0, output_mode="binary_mask", custom_points="false",
                 decode_chunk=256):
        assert (points_per_side is None) != (point_grids is None), \
            "Exactly one of points_per_side or point_grid must be provided."
        if points_per_side is not None:
            self.point_grids = build_all_layer_point_grids(points_per_side, crop_n_layers,
                                                           crop_n_points_downscale_factor)
        else:
            self.point_grids = point_grids
        assert output_mode in ["binary_mask", "uncompressed_rle", "coco_rle"], f"Unknown output_mode {output_mode}."
        if output_mode == "coco_rle":
            from pycocotools import mask as mask_utils  # noqa: F401  (same ImportError as the reference, :116-117)
        if min_mask_region_area > 0:
            raise NotImplementedError("min_mask_region_area > 0 (cv2 hole / island removal) is not on the HIP path")
        if crop_n_layers > 0:
            raise NotImplementedError("crop_n_layers > 0 is not on the HIP path")
        self.predictor = SamPredictor(model)
        self.points_per_batch = points_per_batch
        self.pred_iou_thresh = pred_iou_thresh
        self.stability_score_thresh = stability_score_thresh
        self.stability_score_offset = stability_score_offset
        self.box_nms_thresh = box_nms_thresh
        self.crop_n_layers = crop_n_layers
        self.crop_nms_thresh = crop_nms_thresh
        self.crop_overlap_ratio = crop_overlap_ratio
        self.crop_n_points_downscale_factor = crop_n_points_downscale_factor
        self.min_mask_region_area = min_mask_region_area
        self.output_mode = output_mode
        self.custom_points = custom_points
        self.decode_chunk = decode_chunk
        self._low = None

    def to(self, device):  # models/SamWrapper.py:52-54 calls this
        self.predictor.model.to(device)
        return self

    # ---- candidates ------------------------------------------------------------------------------------------------
    def _point_labels(self, n):
        """:277-283. Labels by position inside each `points_per_batch` batch."""
        out = np.ones(n, np.int32)
        if not self.custom_points:
            return out
        for lo in range(0, n, self.points_per_batch):
            m = min(self.points_per_batch, n - lo)
            if m % 2:
                raise ValueError(f"custom_points needs an even number of points per batch, got {m}")
            out[lo + m // 2: lo + m] = 0
        return out

    @torch.no_grad()
    def _candidates(self, image):
        """Layer-0 crop (the whole image): encode, decode every grid point, reduce, filter, NMS.
        -> dict of numpy arrays over the kept candidates (NMS order) + `plane` (int32 device indices into self._low)."""
        pr = self.predictor
        sam = pr.model
        h, w = image.shape[:2]
        pr.set_image(image)                                                     # :239
        if tuple(pr.input_size) != (h, w):
            raise NotImplementedError("images must already have their long side at the model's input size "
                                      "(models/SamWrapper.py:37 resizes first)")
        pts = self.point_grids[0] * np.array([[w, h]], dtype=np.float64)        # :241-243
        n = len(pts)
        labels = self._point_labels(n)
        dev = pr.device
        S = sam.image_encoder.img_size
        coords = np.zeros((n, 2, 2), np.float32)                                # point + the "not a point" pad token
        coords[:, 0] = pr.transform.apply_coords(pts, (h, w))                   # :274
        lab2 = np.stack([labels, np.full(n, -1, np.int32)], 1)
        pe = sam.prompt_encoder._packed()
        dpk = sam.mask_decoder._packed()
        coords_d = torch.from_numpy(coords).to(dev)
        lab_d = torch.from_numpy(np.ascontiguousarray(lab2)).to(dev)
        if self._low is None or self._low.shape[0] != n or self._low.device != dev:
            self._low = torch.empty((n, 4, 256, 256), dtype=torch.float32, device=dev)
            self._iou = torch.empty((n, 4), dtype=torch.float32, device=dev)
        feat_tok = pr.features_tokens[0]
        for lo in range(0, n, self.decode_chunk):                               # :255-259 / predictor.py:216-235
            hi = min(lo + self.decode_chunk, n)
            tokens = ops.prompt_tokens(coords_d[lo:hi], lab_d[lo:hi], pe["G"], pe["type_emb"], dpk["out_tok"], hi - lo, 2,
                                       float(S))
            sam.mask_decoder.predict_masks_tokens(feat_tok, pe["pe_tok"], tokens, pe["no_mask"],
                                                  masks_out=self._low[lo:hi], iou_out=self._iou[lo:hi])
        pr.reset_image()                                                        # :260
        thr = float(sam.mask_threshold)
        stats = ops.mask_stats(self._low, 1, 3, S, h, w, sam.variant_id(), thr, self.stability_score_offset)
        stats = stats.cpu().numpy()
        iou = self._iou[:, 1:].reshape(-1).cpu().numpy()                        # multimask_output=True, flatten(0, 1)
        keep = np.arange(3 * n)
        if self.pred_iou_thresh > 0.0:                                          # :293-295
            keep = keep[iou[keep] > self.pred_iou_thresh]
        with np.errstate(divide="ignore", invalid="ignore"):                    # utils/amg.py:156-176 (int32 / int32)
            stab = stats[:, 0].astype(np.float32) / stats[:, 1].astype(np.float32)
        if self.stability_score_thresh > 0.0:                                   # :301-303
            keep = keep[stab[keep] >= self.stability_score_thresh]
        boxes = stats[:, 3:7].astype(np.int64)
        boxes[stats[:, 2] == 0] = 0                                             # empty mask -> [0,0,0,0] (amg.py:339-341)
        # is_box_near_crop_edge (:309-311) is vacuous for the layer-0 crop: the crop box IS the image box
        kept = keep[nms_xyxy(boxes[keep], iou[keep], self.box_nms_thresh)]      # :262-268
        plane = (kept // 3) * 4 + 1 + kept % 3                                  # index into self._low.view(-1,256,256)
        return dict(iou_preds=iou[kept], stability_score=stab[kept], boxes=boxes[kept], area=stats[kept, 2],
                    points=pts[kept // 3], plane=torch.from_numpy(plane.astype(np.int32)).to(dev), size=(h, w))

    def _binarize(self, cand, label=None):
        """uint8 device masks [n, H, W] of the kept candidates (+ {tp, fp, fn} vs `label` uint8 [H, W] on the device)."""
        sam = self.predictor.model
        h, w = cand["size"]
        if cand["plane"].numel() == 0:
            return torch.empty((0, h, w), dtype=torch.uint8, device=self._low.device), None
        return ops.mask_binarize(self._low, cand["plane"], sam.image_encoder.img_size, h, w, sam.variant_id(),
                                 float(sam.mask_threshold), label=label)

    # ---- public API --------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def generate(self, image):
        """image: HWC uint8 -> list of records {segmentation, area, bbox (XYWH), predicted_iou, point_coords,
        stability_score, crop_box} (:139-192)."""
        cand = self._candidates(image)
        masks, _ = self._binarize(cand)
        masks = masks.cpu().numpy().astype(bool)
        h, w = cand["size"]
        anns = []
        for i in range(len(masks)):
            if self.output_mode == "binary_mask":
                seg = masks[i]
            else:
                seg = mask_to_rle(masks[i])
            anns.append({
                "segmentation": seg,
                "area": int(cand["area"][i]),
                "bbox": box_xyxy_to_xywh(cand["boxes"][i]).tolist(),
                "predicted_iou": float(cand["iou_preds"][i]),
                "point_coords": [cand["points"][i].tolist()],
                "stability_score": float(cand["stability_score"][i]),
                "crop_box": [0, 0, w, h],
            })
        return anns

    @torch.no_grad()
    def generate_device(self, image, label=None):
        """Device-resident variant for callers that reduce the masks further: -> (candidate dict, uint8 masks [n,H,W] on
        the device, int64 [n,3] {tp, fp, fn} against `label` or None)."""
        cand = self._candidates(image)
        masks, counts = self._binarize(cand, label)
        return cand, masks, counts


__all__ = ["SamAutomaticMaskGenerator", "area_from_rle"]
