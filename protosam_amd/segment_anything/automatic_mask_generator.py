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

Crop layers (`crop_n_layers > 0`, :194-262), images whose long side is not the model's input size, and
`min_mask_region_area > 0` (:332-380) take the general path (`_generate_general`): every crop is re-encoded; its candidates
take the SECOND resize of `postprocess_masks`, so the ones that pass the predicted-IoU filter are materialised at the crop's
resolution in chunks and reduced by `psam_plane_stats` (counts, box, binary mask in one pass); edge filter, per-crop and
cross-crop NMS on the host over the statistics; holes / islands below the area threshold are found with the device
connected-components kernel (`psam_ccl`; the reference loops over masks with cv2, utils/amg.py:267-291).
Not on this path (raise): output_mode "coco_rle" (pycocotools, same ImportError as the reference).
"""
import numpy as np
import torch

from .. import ops
from .predictor import SamPredictor
from .utils.amg import (box_xyxy_to_xywh, build_all_layer_point_grids, generate_crop_boxes,
                        is_box_near_crop_edge, mask_to_rle, nms_xyxy)


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
        assert tuple(pr.input_size) == (h, w)                                   # (other sizes take _generate_general)
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

    # ---- general path: crop layers, any image size, small-region removal -----------------------------------------------------
    def _fast_path(self, image):
        """Layer-0 crop only, image already at the model's input size (what SamWrapper.forward produces): the fused path."""
        h, w = image.shape[:2]
        S = self.predictor.model.image_encoder.img_size
        return self.crop_n_layers == 0 and self.min_mask_region_area == 0 and max(h, w) == S

    def _decode_points(self, pts, im_size):
        """All grid points of the image set on the predictor -> self._low [n,4,256,256], self._iou [n,4] (device)."""
        pr = self.predictor
        sam = pr.model
        n = len(pts)
        dev = pr.device
        S = sam.image_encoder.img_size
        labels = self._point_labels(n)
        coords = np.zeros((n, 2, 2), np.float32)
        coords[:, 0] = pr.transform.apply_coords(pts, im_size)
        lab2 = np.stack([labels, np.full(n, -1, np.int32)], 1)
        pe = sam.prompt_encoder._packed()
        dpk = sam.mask_decoder._packed()
        coords_d = torch.from_numpy(coords).to(dev)
        lab_d = torch.from_numpy(np.ascontiguousarray(lab2)).to(dev)
        if self._low is None or self._low.shape[0] != n or self._low.device != dev:
            self._low = torch.empty((n, 4, 256, 256), dtype=torch.float32, device=dev)
            self._iou = torch.empty((n, 4), dtype=torch.float32, device=dev)
        feat_tok = pr.features_tokens[0]
        for lo in range(0, n, self.decode_chunk):
            hi = min(lo + self.decode_chunk, n)
            tokens = ops.prompt_tokens(coords_d[lo:hi], lab_d[lo:hi], pe["G"], pe["type_emb"], dpk["out_tok"], hi - lo, 2,
                                       float(S))
            sam.mask_decoder.predict_masks_tokens(feat_tok, pe["pe_tok"], tokens, pe["no_mask"],
                                                  masks_out=self._low[lo:hi], iou_out=self._iou[lo:hi])

    def _process_crop(self, image, crop_box, layer_idx, orig_size, chunk=64):
        """:221-316 for one crop. -> dict of host arrays over the candidates kept after the per-crop NMS, in the ORIGINAL
        image's frame, + `masks` uint8 [k, H, W] on the device (uncropped)."""
        pr = self.predictor
        sam = pr.model
        H, W = orig_size
        x0, y0, x1, y1 = crop_box
        cropped = np.ascontiguousarray(image[y0:y1, x0:x1, :])
        ch, cw = cropped.shape[:2]
        pr.set_image(cropped)
        pts = self.point_grids[layer_idx] * np.array([[cw, ch]], dtype=np.float64)
        self._decode_points(pts, (ch, cw))
        in_size = tuple(pr.input_size)
        pr.reset_image()
        dev = self._low.device
        thr = float(sam.mask_threshold)
        n = len(pts)
        iou = self._iou[:, 1:].reshape(-1).cpu().numpy()
        cand = np.arange(3 * n)
        if self.pred_iou_thresh > 0.0:
            cand = cand[iou[cand] > self.pred_iou_thresh]
        planes = (cand // 3) * 4 + 1 + cand % 3
        low_flat = self._low.view(-1, 1, 256, 256)
        keep_l, stab_l, box_l, area_l, mask_l = [], [], [], [], []
        for lo in range(0, len(cand), chunk):                                   # bounded full-resolution working set
            sel = torch.from_numpy(planes[lo:lo + chunk].astype(np.int64)).to(dev)
            full = sam.postprocess_masks(low_flat[sel].contiguous(), in_size, (ch, cw))[:, 0].contiguous()   # [k, ch, cw] logits
            st, binm = ops.plane_stats(full, thr, self.stability_score_offset)
            st = st.cpu().numpy()
            with np.errstate(divide="ignore", invalid="ignore"):
                stab = st[:, 0].astype(np.float32) / st[:, 1].astype(np.float32)
            ok = np.ones(len(st), bool)
            if self.stability_score_thresh > 0.0:
                ok &= stab >= self.stability_score_thresh
            boxes = st[:, 3:7].astype(np.int64)
            boxes[st[:, 2] == 0] = 0
            ok &= ~is_box_near_crop_edge(boxes, crop_box, [0, 0, W, H])         # :309-311
            idx = np.flatnonzero(ok)
            keep_l.append(cand[lo:lo + chunk][idx]); stab_l.append(stab[idx]); box_l.append(boxes[idx]); area_l.append(st[idx, 2])
            mask_l.append(binm[torch.from_numpy(idx).to(dev)])
        kept = np.concatenate(keep_l) if keep_l else np.zeros(0, np.int64)
        stab = np.concatenate(stab_l) if stab_l else np.zeros(0, np.float32)
        boxes = np.concatenate(box_l) if box_l else np.zeros((0, 4), np.int64)
        area = np.concatenate(area_l) if area_l else np.zeros(0, np.int64)
        masks = torch.cat(mask_l) if mask_l else torch.empty((0, ch, cw), dtype=torch.uint8, device=dev)
        order = nms_xyxy(boxes, iou[kept], self.box_nms_thresh)                 # :244-250
        full_masks = torch.zeros((len(order), H, W), dtype=torch.uint8, device=dev)   # uncrop_masks
        if len(order):
            full_masks[:, y0:y1, x0:x1] = masks[torch.from_numpy(order).to(dev)]
        return dict(iou_preds=iou[kept][order], stability_score=stab[order], boxes=boxes[order] + np.array([[x0, y0, x0, y0]]),
                    area=area[order], points=pts[kept[order] // 3] + np.array([[x0, y0]], dtype=np.float64),
                    crop_boxes=np.tile(np.array([crop_box], dtype=np.int64), (len(order), 1)), masks=full_masks)

    def _zero_plane(self, H, W, dev):
        z = getattr(self, "_zero_planes", None)
        if z is None:
            z = self._zero_planes = {}
        key = (H, W, str(dev))
        if key not in z:
            z.clear()
            z[key] = torch.zeros((H, W), dtype=torch.float32, device=dev)
        return z[key]

    def _remove_small_regions(self, mask_u8, area_thresh, mode, cw):
        """utils/amg.py:267-291 on the device: connected components of the mask (islands) or of its complement (holes) with
        `psam_ccl`, areas from its table, relabelling through a small lookup table. -> (uint8 mask [H,W], changed)."""
        H, W = mask_u8.shape
        holes = mode == "holes"
        working = (1 - mask_u8) if holes else mask_u8
        ops.ccl(working.contiguous(), self._zero_plane(H, W, mask_u8.device), cw)
        tab = cw.tab.cpu().numpy()
        if int(tab[0]) > int(tab[1]):
            # more regions than the table has rows (speckled crop-resolution logits; cv2.connectedComponentsWithStats has no
            # limit): the union-find forest of the same labelling is complete - every foreground pixel of `working` holds its
            # region's root pixel - so the areas come from it; roots ascend like the table's rows, so ties break alike
            par = cw.parent.view(H, W)
            fgm = par >= 0
            _, inv, counts = torch.unique(par[fgm], return_inverse=True, return_counts=True)
            small_t = counts < area_thresh
            if not bool(small_t.any()):
                return mask_u8, False
            sel = torch.zeros((H, W), dtype=torch.uint8, device=mask_u8.device)
            if holes:
                sel[fgm] = small_t[inv].to(torch.uint8)
                return mask_u8 | sel, True
            keep = ~small_t
            if not bool(keep.any()):
                keep[int(counts.argmax())] = True
            sel[fgm] = keep[inv].to(torch.uint8)
            return sel, True
        n = int(tab[1])
        sizes = tab[ops.CC_HDR:ops.CC_HDR + ops.CC_STRIDE * n].reshape(n, ops.CC_STRIDE)[:, 0]
        small = [i + 1 for i, sz in enumerate(sizes) if sz < area_thresh]
        if len(small) == 0:
            return mask_u8, False
        fill = [0] + small
        if not holes:
            fill = [i for i in range(n + 1) if i not in fill]
            if len(fill) == 0:
                fill = [int(np.argmax(sizes)) + 1]
        lut = np.zeros(n + 1, np.uint8)
        lut[fill] = 1
        new = torch.from_numpy(lut).to(mask_u8.device)[cw.labels.view(H, W).long()]
        return new, True

    @torch.no_grad()
    def _generate_general(self, image):
        """-> (dict of host arrays, uint8 masks [m, H, W] on the device) over the final records."""
        H, W = image.shape[:2]
        crop_boxes, layer_idxs = generate_crop_boxes((H, W), self.crop_n_layers, self.crop_overlap_ratio)
        parts = [self._process_crop(image, cb, li, (H, W)) for cb, li in zip(crop_boxes, layer_idxs)]
        data = {k: np.concatenate([p[k] for p in parts]) for k in parts[0] if k != "masks"}
        masks = torch.cat([p["masks"] for p in parts])
        dev = masks.device
        if len(crop_boxes) > 1:                                                 # :208-218 prefer masks from smaller crops
            cb = data["crop_boxes"].astype(np.float32)
            scores = 1.0 / ((cb[:, 2] - cb[:, 0]) * (cb[:, 3] - cb[:, 1]))
            keep = nms_xyxy(data["boxes"], scores, self.crop_nms_thresh)
            data = {k: v[keep] for k, v in data.items()}
            masks = masks[torch.from_numpy(keep).to(dev)]
        if self.min_mask_region_area > 0 and len(masks):                        # :332-380
            cw = ops.CclWorkspace(H, W, 4096, dev)
            new_masks, scores = [], []
            for i in range(len(masks)):
                m, c1 = self._remove_small_regions(masks[i], self.min_mask_region_area, "holes", cw)
                m, c2 = self._remove_small_regions(m, self.min_mask_region_area, "islands", cw)
                new_masks.append(m)
                scores.append(float(not c1 and not c2))
            nm = torch.stack(new_masks)
            st, _ = ops.plane_stats(nm.float().contiguous(), 0.5, 0.0, binarize=False)
            st = st.cpu().numpy()
            nb = st[:, 3:7].astype(np.int64)
            nb[st[:, 2] == 0] = 0
            keep = nms_xyxy(nb, np.asarray(scores, dtype=np.float32), max(self.box_nms_thresh, self.crop_nms_thresh))
            for i in keep:
                if scores[i] == 0.0:
                    masks[i] = nm[i]
                    data["boxes"][i] = nb[i]
                    data["area"][i] = st[i, 2]
            data = {k: v[keep] for k, v in data.items()}
            masks = masks[torch.from_numpy(keep).to(dev)]
        return data, masks

    # ---- public API --------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def generate(self, image):
        """image: HWC uint8 -> list of records {segmentation, area, bbox (XYWH), predicted_iou, point_coords,
        stability_score, crop_box} (:139-192)."""
        if not self._fast_path(image):
            data, masks = self._generate_general(image)
            masks = masks.cpu().numpy().astype(bool)
            anns = []
            for i in range(len(masks)):
                anns.append({
                    "segmentation": masks[i] if self.output_mode == "binary_mask" else mask_to_rle(masks[i]),
                    "area": int(data["area"][i]),
                    "bbox": box_xyxy_to_xywh(data["boxes"][i]).tolist(),
                    "predicted_iou": float(data["iou_preds"][i]),
                    "point_coords": [data["points"][i].tolist()],
                    "stability_score": float(data["stability_score"][i]),
                    "crop_box": box_xyxy_to_xywh(data["crop_boxes"][i]).tolist(),
                })
            return anns
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
        if not self._fast_path(image):
            data, masks = self._generate_general(image)
            counts = None
            if label is not None:   # {tp, fp, fn} of every record against the label (models/SamWrapper.py:8-13), on the device
                lab = (label != 0)
                m = masks != 0
                counts = torch.stack([(m & lab).flatten(1).sum(1), (m & ~lab).flatten(1).sum(1), (~m & lab).flatten(1).sum(1)], 1)
            data = dict(data, size=tuple(image.shape[:2]))
            return data, masks, counts
        cand = self._candidates(image)
        masks, counts = self._binarize(cand, label)
        return cand, masks, counts


__all__ = ["SamAutomaticMaskGenerator", "area_from_rle"]
