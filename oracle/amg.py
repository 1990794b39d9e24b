"""Oracle: SamAutomaticMaskGenerator.generate and SamWrapper.forward (numpy / fp32 CPU torch).
Test infrastructure only (see oracle/__init__.py).

Follows models/segment_anything/automatic_mask_generator.py (generate :139-192, _generate_masks :194-219,
_process_crop :221-262, _process_batch :264-316, postprocess_small_regions :332-380) including crop layers
(crop_n_layers > 0: generate_crop_boxes utils/amg.py:202-237, is_box_near_crop_edge :78-88, uncrop_* :240-265) and
min_mask_region_area > 0 (remove_small_regions :267-291, on the restated connected components of oracle/glue.py - cv2 is
absent), models/segment_anything/utils/amg.py (build_point_grid :179-187,
calculate_stability_score :156-176, batched_mask_to_box :303-346, mask_to_rle_pytorch / rle_to_mask :108-153) and
models/SamWrapper.py (get_iou :8-13, forward :29-50).

`torchvision.ops.boxes.batched_nms` (torchvision is absent here and from /root/reference) => PARITY UNPINNED for that one
call; `nms` below restates its published contract (all categories are 0 => plain NMS: visit by decreasing score, drop
later boxes whose IoU with a kept box is > threshold, area = (x2-x1)*(y2-y1)). oracle/validate_against_reference.py
injects this same function where the reference imports torchvision's, so everything around it is pinned.
"""
import numpy as np
import torch

from . import sam_image_encoder as oenc
from . import sam_prompt_decoder as odec
from .glue import sam_preprocess


def point_grid(n):
    """utils/amg.py:179-187."""
    off = 1 / (2 * n)
    side = np.linspace(off, 1 - off, n)
    return np.stack([np.tile(side[None, :], (n, 1)), np.tile(side[:, None], (1, n))], axis=-1).reshape(-1, 2)


def nms(boxes, scores, thr):
    boxes = boxes.float()
    order = torch.argsort(scores.float(), descending=True, stable=True).tolist()
    area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    keep, dead = [], [False] * len(order)
    for a, i in enumerate(order):
        if dead[i]:
            continue
        keep.append(i)
        for j in order[a + 1:]:
            if dead[j]:
                continue
            w = max(0.0, float(min(boxes[i, 2], boxes[j, 2]) - max(boxes[i, 0], boxes[j, 0])))
            h = max(0.0, float(min(boxes[i, 3], boxes[j, 3]) - max(boxes[i, 1], boxes[j, 1])))
            inter = np.float32(w) * np.float32(h)
            with np.errstate(divide="ignore", invalid="ignore"):
                iou = inter / (np.float32(area[i]) + np.float32(area[j]) - inter)
            if float(iou) > thr:
                dead[j] = True
    return torch.tensor(keep, dtype=torch.long)


def batched_nms(boxes, scores, idxs, iou_threshold):
    """torchvision.ops.boxes.batched_nms for the generator's use (idxs all zero)."""
    assert int(torch.count_nonzero(idxs)) == 0
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.long)
    return nms(boxes, scores, iou_threshold)


def mask_to_box(masks):
    """utils/amg.py:303-346 for bool [N,H,W] -> int64 [N,4] XYXY, zeros for an empty mask."""
    out = torch.zeros((masks.shape[0], 4), dtype=torch.long)
    for i, m in enumerate(masks):
        ys, xs = torch.nonzero(m, as_tuple=True)
        if len(ys):
            out[i] = torch.tensor([xs.min(), ys.min(), xs.max(), ys.max()])
    return out


def rle(mask):
    """utils/amg.py:108-134 for one bool [H,W] mask."""
    h, w = mask.shape
    flat = np.asarray(mask).T.reshape(-1)
    idx = np.flatnonzero(flat[1:] ^ flat[:-1])
    cur = np.concatenate([[0], idx + 1, [h * w]])
    counts = [] if flat[0] == 0 else [0]
    counts.extend((cur[1:] - cur[:-1]).tolist())
    return {"size": [h, w], "counts": counts}


def rle_to_mask(r):
    """utils/amg.py:137-149."""
    h, w = r["size"]
    m = np.empty(h * w, dtype=bool)
    i, parity = 0, False
    for c in r["counts"]:
        m[i:i + c] = parity
        i += c
        parity ^= True
    return m.reshape(w, h).transpose()


def crop_boxes_for(im_size, n_layers, overlap_ratio):
    """utils/amg.py:202-237 generate_crop_boxes -> (boxes XYXY, layer index per box)."""
    import math
    from itertools import product
    boxes, layers = [], []
    im_h, im_w = im_size
    short = min(im_h, im_w)
    boxes.append([0, 0, im_w, im_h])
    layers.append(0)
    for i_layer in range(n_layers):
        n = 2 ** (i_layer + 1)
        overlap = int(overlap_ratio * short * (2 / n))
        cw = int(math.ceil((overlap * (n - 1) + im_w) / n))
        ch = int(math.ceil((overlap * (n - 1) + im_h) / n))
        for x0, y0 in product([int((cw - overlap) * i) for i in range(n)], [int((ch - overlap) * i) for i in range(n)]):
            boxes.append([x0, y0, min(x0 + cw, im_w), min(y0 + ch, im_h)])
            layers.append(i_layer + 1)
    return boxes, layers


def box_near_crop_edge(boxes, crop_box, orig_box, atol=20.0):
    """utils/amg.py:78-88 (boxes int64 [n,4] in the crop's frame)."""
    cb = torch.as_tensor(crop_box, dtype=torch.float)
    ob = torch.as_tensor(orig_box, dtype=torch.float)
    b = (boxes + torch.tensor([[crop_box[0], crop_box[1], crop_box[0], crop_box[1]]])).float()
    near_crop = torch.isclose(b, cb[None, :], atol=atol, rtol=0)
    near_img = torch.isclose(b, ob[None, :], atol=atol, rtol=0)
    return torch.any(near_crop & ~near_img, dim=1)


def remove_small_regions(mask, area_thresh, mode):
    """utils/amg.py:267-291 with cv2.connectedComponentsWithStats(., 8) restated (oracle/glue.py)."""
    from .glue import connected_components_with_stats
    correct_holes = mode == "holes"
    working = (correct_holes ^ mask).astype(np.uint8)
    n_labels, regions, stats, _ = connected_components_with_stats(working)
    sizes = stats[:, -1][1:]
    small = [i + 1 for i, s_ in enumerate(sizes) if s_ < area_thresh]
    if len(small) == 0:
        return mask, False
    fill = [0] + small
    if not correct_holes:
        fill = [i for i in range(n_labels) if i not in fill]
        if len(fill) == 0:
            fill = [int(np.argmax(sizes)) + 1]
    return np.isin(regions, fill), True


def generate(image_u8, sd, sam_type="vit_b", points_per_side=32, points_per_batch=64, pred_iou_thresh=0.88,
             stability_score_thresh=0.95, stability_score_offset=1.0, box_nms_thresh=0.7, custom_points="false",
             postprocess="batched", encoder_depth=None, features=None, taps=None, crop_n_layers=0, crop_nms_thresh=0.7,
             crop_overlap_ratio=512 / 1500, crop_n_points_downscale_factor=1, min_mask_region_area=0):
    """-> list of records as SamAutomaticMaskGenerator.generate (output_mode 'binary_mask')."""
    if crop_n_layers == 0 and min_mask_region_area == 0:
        return _generate_layer0(image_u8, sd, sam_type, points_per_side, points_per_batch, pred_iou_thresh,
                                stability_score_thresh, stability_score_offset, box_nms_thresh, custom_points, postprocess,
                                encoder_depth, features, taps)
    from .glue import apply_image
    H, W = image_u8.shape[:2]
    boxes, layers = crop_boxes_for((H, W), crop_n_layers, crop_overlap_ratio)
    grids = [point_grid(int(points_per_side / (crop_n_points_downscale_factor ** i))) for i in range(crop_n_layers + 1)]
    pe = odec.dense_pe(sd)
    D = dict(iou=[], stab=[], box=[], pts=[], mask=[], crop=[])
    for cb, li in zip(boxes, layers):                                           # :204-206
        x0, y0, x1, y1 = cb
        crop = image_u8[y0:y1, x0:x1, :]
        ch, cw = crop.shape[:2]
        rz = apply_image(crop)                                                  # set_image: PIL resize of the long side
        in_size = tuple(rz.shape[:2])
        feats = oenc.image_encoder(sam_preprocess(rz), sd, model_type=sam_type, depth=encoder_depth)
        pts_all = grids[li] * np.array([[cw, ch]])
        c = dict(iou=[], stab=[], box=[], pts=[], mask=[])
        for lo in range(0, len(pts_all), points_per_batch):
            pts = pts_all[lo:lo + points_per_batch]
            tp = torch.as_tensor(odec.apply_coords(pts, (ch, cw)))
            if custom_points:
                pos = torch.ones(tp.shape[0] // 2, dtype=torch.int)
                labels = torch.cat((pos, torch.zeros_like(pos)), dim=0)
            else:
                labels = torch.ones(tp.shape[0], dtype=torch.int)
            sparse, dense = odec.prompt_encoder(sd, (tp[:, None, :], labels[:, None]), None)
            low, iou = odec.mask_decoder(sd, feats, pe, sparse, dense, True)
            masks = odec.postprocess_masks(low, in_size, (ch, cw), postprocess)
            masks, iou = masks.flatten(0, 1), iou.flatten(0, 1)
            bp = torch.as_tensor(pts.repeat(3, axis=0))
            keep = torch.ones(len(iou), dtype=torch.bool)
            if pred_iou_thresh > 0.0:
                keep &= iou > pred_iou_thresh
            masks, iou, bp = masks[keep], iou[keep], bp[keep]
            inter = (masks > (0.0 + stability_score_offset)).sum(-1, dtype=torch.int16).sum(-1, dtype=torch.int32)
            union = (masks > (0.0 - stability_score_offset)).sum(-1, dtype=torch.int16).sum(-1, dtype=torch.int32)
            stab = inter / union
            if stability_score_thresh > 0.0:
                k2 = stab >= stability_score_thresh
                masks, iou, bp, stab = masks[k2], iou[k2], bp[k2], stab[k2]
            m = masks > 0.0
            bx = mask_to_box(m)
            k3 = ~box_near_crop_edge(bx, cb, [0, 0, W, H])                      # :309-311
            m, iou, bp, stab, bx = m[k3], iou[k3], bp[k3], stab[k3], bx[k3]
            full = torch.zeros((m.shape[0], H, W), dtype=torch.bool)            # uncrop_masks :252-265
            full[:, y0:y1, x0:x1] = m
            c["iou"].append(iou); c["stab"].append(stab); c["box"].append(bx); c["pts"].append(bp); c["mask"].append(full)
        ci = {k: torch.cat(v) for k, v in c.items()}
        keep = batched_nms(ci["box"].float(), ci["iou"], torch.zeros_like(ci["box"][:, 0]), box_nms_thresh)   # :244-250
        D["iou"].append(ci["iou"][keep]); D["stab"].append(ci["stab"][keep]); D["mask"].append(ci["mask"][keep])
        D["box"].append(ci["box"][keep] + torch.tensor([[x0, y0, x0, y0]]))     # uncrop_boxes_xyxy
        D["pts"].append(ci["pts"][keep] + torch.tensor([[x0, y0]]))             # uncrop_points
        D["crop"].append(torch.tensor([cb for _ in range(len(keep))]).reshape(-1, 4))
    A = {k: torch.cat(v) for k, v in D.items()}
    if len(boxes) > 1:                                                          # :208-218: prefer masks from smaller crops
        cr = A["crop"].float()
        scores = 1 / ((cr[:, 2] - cr[:, 0]) * (cr[:, 3] - cr[:, 1]))
        keep = batched_nms(A["box"].float(), scores, torch.zeros_like(A["box"][:, 0]), crop_nms_thresh)
        A = {k: v[keep] for k, v in A.items()}
    masks_np = [rle_to_mask(rle(m.numpy())) for m in A["mask"]]
    boxes_t = A["box"].clone()
    keep_final = list(range(len(masks_np)))
    if min_mask_region_area > 0 and len(masks_np):                              # :332-380
        new_masks, scores = [], []
        for m in masks_np:
            m2, ch1 = remove_small_regions(m, min_mask_region_area, "holes")
            m2, ch2 = remove_small_regions(m2, min_mask_region_area, "islands")
            new_masks.append(torch.as_tensor(m2).unsqueeze(0))
            scores.append(float(not ch1 and not ch2))
        nm = torch.cat(new_masks, dim=0)
        nb = mask_to_box(nm)
        keep = batched_nms(nb.float(), torch.as_tensor(scores), torch.zeros_like(nb[:, 0]),
                           max(box_nms_thresh, crop_nms_thresh))
        for i in keep.tolist():
            if scores[i] == 0.0:
                masks_np[i] = rle_to_mask(rle(nm[i].numpy()))
                boxes_t[i] = nb[i]
        keep_final = keep.tolist()
    anns = []
    for i in keep_final:
        b = boxes_t[i].tolist()
        cbx = A["crop"][i].tolist()
        anns.append({"segmentation": masks_np[i], "area": int(masks_np[i].sum()), "bbox": [b[0], b[1], b[2] - b[0], b[3] - b[1]],
                     "predicted_iou": A["iou"][i].item(), "point_coords": [A["pts"][i].tolist()],
                     "stability_score": A["stab"][i].item(),
                     "crop_box": [cbx[0], cbx[1], cbx[2] - cbx[0], cbx[3] - cbx[1]]})
    return anns


def _generate_layer0(image_u8, sd, sam_type, points_per_side, points_per_batch, pred_iou_thresh, stability_score_thresh,
                     stability_score_offset, box_nms_thresh, custom_points, postprocess, encoder_depth, features, taps):
    """The layer-0 crop only (crop_n_layers = 0, min_mask_region_area = 0: the reference's defaults), for an image whose long
    side is already 1024 and square (SamWrapper.forward resizes before calling, SamWrapper.py:37)."""
    h, w = image_u8.shape[:2]
    assert (h, w) == (1024, 1024)
    if features is None:
        features = oenc.image_encoder(sam_preprocess(image_u8), sd, model_type=sam_type, depth=encoder_depth)  # set_image
    pts_all = point_grid(points_per_side) * np.array([[w, h]])                  # :241-243
    pe = odec.dense_pe(sd)
    iou_l, stab_l, box_l, pts_l, mask_l = [], [], [], [], []
    all_iou, all_stab = [], []
    for lo in range(0, len(pts_all), points_per_batch):                         # :246-249
        pts = pts_all[lo:lo + points_per_batch]
        tp = torch.as_tensor(odec.apply_coords(pts, (h, w)))                    # float64, :274-275
        if custom_points:                                                       # :277-283 (the default "false" is truthy)
            pos = torch.ones(tp.shape[0] // 2, dtype=torch.int)
            labels = torch.cat((pos, torch.zeros_like(pos)), dim=0)
        else:
            labels = torch.ones(tp.shape[0], dtype=torch.int)
        sparse, dense = odec.prompt_encoder(sd, (tp[:, None, :], labels[:, None]), None)
        low, iou = odec.mask_decoder(sd, features, pe, sparse, dense, True)     # predictor.py:229-235
        masks = odec.postprocess_masks(low, (h, w), (h, w), postprocess)        # return_logits=True
        masks, iou = masks.flatten(0, 1), iou.flatten(0, 1)                     # :288-292
        bp = torch.as_tensor(pts.repeat(3, axis=0))
        inter = (masks > (0.0 + stability_score_offset)).sum(-1, dtype=torch.int16).sum(-1, dtype=torch.int32)
        union = (masks > (0.0 - stability_score_offset)).sum(-1, dtype=torch.int16).sum(-1, dtype=torch.int32)
        stab_full = inter / union
        all_iou.append(iou.clone())
        all_stab.append(stab_full.clone())
        keep = torch.ones(len(iou), dtype=torch.bool)
        if pred_iou_thresh > 0.0:                                               # :295-297
            keep &= iou > pred_iou_thresh
        if stability_score_thresh > 0.0:                                        # :303-305
            keep &= stab_full >= stability_score_thresh
        m = masks[keep] > 0.0                                                   # :308
        iou_l.append(iou[keep]); stab_l.append(stab_full[keep]); pts_l.append(bp[keep])
        box_l.append(mask_to_box(m)); mask_l.append(m)                          # :309; edge filter vacuous for crop == image
    iou_c, stab_c, pts_c = torch.cat(iou_l), torch.cat(stab_l), torch.cat(pts_l)
    box_c, mask_c = torch.cat(box_l), torch.cat(mask_l)
    if taps is not None:
        taps.update(iou_all=torch.cat(all_iou), stab_all=torch.cat(all_stab), features=features)
    keep = batched_nms(box_c.float(), iou_c, torch.zeros_like(box_c[:, 0]), box_nms_thresh)   # :262-268
    anns = []
    for i in keep.tolist():
        b = box_c[i].tolist()
        seg = rle_to_mask(rle(mask_c[i].numpy()))                               # :312-313, :181
        anns.append({"segmentation": seg, "area": int(seg.sum()), "bbox": [b[0], b[1], b[2] - b[0], b[3] - b[1]],
                     "predicted_iou": iou_c[i].item(), "point_coords": [pts_c[i].tolist()],
                     "stability_score": stab_c[i].item(), "crop_box": [0, 0, w, h]})
    return anns


def get_iou(mask, label):
    """models/SamWrapper.py:8-13."""
    tp = (mask * label).sum()
    fp = (mask * (1 - label)).sum()
    fn = ((1 - mask) * label).sum()
    return tp / (tp + fp + fn)


def sam_wrapper_forward(image_u8, image_labels, sd, **kw):
    """models/SamWrapper.py:29-50 for a 1024x1024 image (apply_image is the identity). -> (bool [H,W], index, ious)."""
    masks = generate(image_u8, sd, **kw)
    best, best_iou, ious = None, 0, []
    for i, m in enumerate(masks):
        with np.errstate(divide="ignore", invalid="ignore"):
            iou = get_iou(m["segmentation"].astype(np.uint8), image_labels)
        ious.append(iou)
        if best is None or iou > best_iou:
            best, best_iou = i, iou
    return masks[best]["segmentation"], best, ious, masks
