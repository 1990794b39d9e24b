"""Oracle: host glue of ProtoSAM.forward between the coarse model and SAM (numpy / fp32 CPU).
Test infrastructure only (see oracle/__init__.py).

Follows models/ProtoSAM.py (forward :536-678, get_bbox_per_cc :242-264, get_most_conf_points :266-289,
get_sam_input_points :349-450, predict_w_points_bbox :500-533) and util/utils.py (get_connected_components
:474-494, cca :496-541, get_confidence_from_logits :429-434).

`cv2.connectedComponentsWithStats(..., connectivity=8)` (opencv-python 4.10.0.84) is absent here and on the GPU
box => PARITY UNPINNED against cv2; `connected_components_with_stats` below restates its published contract
(labels image, stats rows [left, top, width, height, area], float64 centroids (x, y), label 0 = background) with
labels numbered by raster order of each component's first pixel, and is cross-checked against scipy.ndimage.
All downstream results are invariant to the label numbering (union of per-component masks, ProtoSAM.py:669).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import sam_image_encoder as oenc
from . import sam_prompt_decoder as odec

PIXEL_MEAN = (123.675, 116.28, 103.53)  # build_sam.py:100
PIXEL_STD = (58.395, 57.12, 57.375)     # build_sam.py:101


def connected_components_with_stats(img):
    """8-connected components of img != 0. Returns (n_labels, labels int32 [H,W], stats int32 [n,5], centroids f64 [n,2])."""
    img = np.asarray(img)
    H, W = img.shape
    fg = np.pad(img != 0, ((0, 0), (1, 1)))
    d = np.diff(fg.astype(np.int8), axis=1)
    rows, starts = np.nonzero(d == 1)
    _, ends = np.nonzero(d == -1)  # exclusive end, same row-major order as starts
    nrun = len(starts)
    parent = np.arange(nrun)

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a

    row_ptr = np.searchsorted(rows, np.arange(H + 1))
    for r in range(1, H):
        a, a_end = row_ptr[r - 1], row_ptr[r]
        b, b_end = row_ptr[r], row_ptr[r + 1]
        while a < a_end and b < b_end:
            # runs touch under 8-connectivity iff a.start <= b.end and b.start <= a.end (ends exclusive)
            if starts[a] <= ends[b] and starts[b] <= ends[a]:
                ra, rb = find(a), find(b)
                if ra != rb:
                    parent[max(ra, rb)] = min(ra, rb)
            if ends[a] < ends[b]:
                a += 1
            else:
                b += 1
    labels = np.zeros((H, W), np.int32)
    ids = {}
    for i in range(nrun):  # runs are in raster order => ids in raster order of each component's first pixel
        root = find(i)
        if root not in ids:
            ids[root] = len(ids) + 1
        labels[rows[i], starts[i]:ends[i]] = ids[root]
    n = len(ids) + 1
    stats = np.zeros((n, 5), np.int32)
    cent = np.zeros((n, 2), np.float64)
    ys, xs = np.mgrid[0:H, 0:W]
    for j in range(n):
        m = labels == j
        area = int(m.sum())
        if area == 0:
            continue
        x0, x1, y0, y1 = xs[m].min(), xs[m].max(), ys[m].min(), ys[m].max()
        stats[j] = (x0, y0, x1 - x0 + 1, y1 - y0 + 1, area)
        cent[j] = (xs[m].sum(dtype=np.float64) / area, ys[m].sum(dtype=np.float64) / area)
    return n, labels, stats, cent


def confidence_from_logits(logits):
    """util/utils.py:429-434."""
    p = logits.softmax(1)[:, 1].flatten(1)
    pred = (p >= 0.5).float()
    return ((p * pred).sum() / (pred.sum() + 1e-6)).item()


def get_connected_components(pred, logits):
    """util/utils.py:474-494. conf[j] = sum(p_fg * [label == j]) / (sum(pred) + 1e-6)."""
    cc = connected_components_with_stats(pred.astype(np.uint8))
    probs = logits.softmax(1)[:, 1].cpu().numpy()
    conf = {0: 0}
    for j in range(1, cc[0]):
        conf[j] = (probs.flatten() * (cc[1] == j).flatten()).sum() / (pred.flatten().sum() + 1e-6)
    return cc, conf


def cca(pred, logits):
    """util/utils.py:496-541 with return_cc=True: keep only the most confident component (relabelled 1)."""
    cc, conf = get_connected_components(pred, logits)
    max_conf, max_key = conf[0], 0
    for k, v in conf.items():
        if v > max_conf:
            max_conf, max_key = v, k
    if max_conf == 0:
        return cc
    return (2, np.where(cc[1] != max_key, 0, 1).astype(np.int32), cc[2][[0, max_key]], cc[3][[0, max_key]])


def cca_pred(pred, logits):
    """util/utils.py:496-541 with return_conf=True: (pred restricted to the most confident component, its confidence);
    all zeros and 0 when there is no component."""
    cc, conf = get_connected_components(pred, logits)
    max_conf, max_key = conf[0], 0
    for k, v in conf.items():
        if v > max_conf:
            max_conf, max_key = v, k
    if max_conf == 0:
        return pred * np.zeros_like(pred), max_conf
    keep = (np.where(cc[1] != max_key, 0, 1) == 1).astype(np.uint8)
    return pred * keep, max_conf


def coarse_pred_only(output_logits, original_size, use_cca):
    """ProtoSAM.py:580-590 / ProtoMedSAM.py:163-172 in eval mode -> (pred int64 [H,W], [conf]). `shape[-2:] != original_size`
    compares a Size with an int there, so the bilinear resize always runs (the identity when the sizes agree)."""
    output_logits = F.interpolate(output_logits, size=original_size, mode="bilinear")
    pred = output_logits.argmax(dim=1)[0]
    conf = confidence_from_logits(output_logits)
    if use_cca:
        _pred, conf = cca_pred(pred.numpy(), output_logits)
        pred = torch.from_numpy(_pred)
    return pred, [conf]


def bbox_per_cc(cc):
    """ProtoSAM.py:242-264: XYXY [min_x, min_y, max_x, max_y] per label >= 1."""
    out = []
    for i in range(1, cc[0]):
        ys, xs = np.nonzero(cc[1] == i)
        out.append([xs.min(), ys.min(), xs.max(), ys.max()])
    return np.array(out)


def most_conf_point(fg_p, comp):
    """ProtoSAM.py:266-289 with k = 1: (x, y) of the largest fg probability inside the component. torch.topk leaves
    tie order unspecified; this oracle (and the HIP path) take the first maximum in raster order."""
    ys, xs = np.nonzero(comp)
    vals = fg_p[ys, xs]
    i = int(np.argmax(vals))
    return np.array([[xs[i], ys[i]]]), [float(vals[i])]


def most_conf_points(fg_p, comp, k):
    """ProtoSAM.get_most_conf_points (ProtoSAM.py:266-289) for k > 1: the component's probabilities in raster order
    (`output_p_fg[mask]`) through torch.topk itself - its order among equal values is the reference's by construction."""
    ys, xs = np.nonzero(comp)
    conf, idx = torch.topk(torch.as_tensor(fg_p[ys, xs]), k)       # (raises for a component of fewer than k pixels, as the reference)
    idx = idx.numpy()
    return np.stack([xs[idx], ys[idx]], axis=1), [float(c) for c in conf]


def dilate3x3(mask, iterations):
    """cv2.dilate(mask, np.ones((3, 3)), iterations=n) on a 0/255 image (cv2 absent: restated; the default border of a
    dilation never contributes). n passes of a 3x3 maximum = one (2n+1) x (2n+1) maximum; done literally here."""
    m = np.asarray(mask) > 0
    for _ in range(iterations):
        p = np.pad(m, 1)
        acc = np.zeros_like(m)
        for dy in range(3):
            for dx in range(3):
                acc |= p[dy:dy + m.shape[0], dx:dx + m.shape[1]]
        m = acc
    return m.astype(np.uint8) * 255


def first_argmax_point(vals_hw, region):
    """get_most_conf_points (ProtoSAM.py:266-289) with k = 1 -> [[x, y]] or None; ties: first in raster order."""
    ys, xs = np.nonzero(region)
    if len(ys) == 0:
        return None
    i = int(np.argmax(vals_hw[ys, xs]))
    return np.array([[xs[i], ys[i]]])


def sam_neg_points(cc, output_p, l=1):
    """ProtoSAM.py:361-372 (global) and :395-419 (per component ring) -> list over components of [n,2] arrays or None."""
    assert l == 1
    bg = output_p[0, 0].cpu().numpy().copy()
    bg_t = bg.copy()
    bg_t[bg_t < 0.95] = 0
    glob = first_argmax_point(bg_t, bg_t > 0)
    out = []
    for cc_id in np.unique(cc[1]):
        if cc_id == 0:
            continue
        pred_u8 = ((cc[1] == cc_id).astype(np.float32) * 255).astype(np.uint8)
        boundary = (dilate3x3(pred_u8, 10) - pred_u8).astype(np.float32) / 255
        neg = first_argmax_point(bg, boundary != 0)
        if neg is not None and glob is not None:
            neg = np.vstack([neg, glob])
        else:
            neg = glob if neg is None else neg
        out.append(neg)
    return out


def sam_input_points(cc, output_p, point_mode="both", k=1):
    """ProtoSAM.py:349-450 (positive points): per component [N,2] points in (x, y); k = num_points_for_sam."""
    fg_p = output_p[0, 1].cpu().numpy()
    pick = most_conf_point if k == 1 else (lambda f, c: most_conf_points(f, c, k))
    pts = []
    for cc_id in np.unique(cc[1]):
        if cc_id == 0:
            continue
        comp = cc[1] == cc_id
        if point_mode == "conf":
            p, _ = pick(fg_p, comp)
        elif point_mode == "centroid":
            p = cc[3][cc_id][None, :]
        elif point_mode == "both":
            p, _ = pick(fg_p, comp)
            p = np.vstack([p, cc[3][cc_id][None, :]])
        else:
            raise NotImplementedError(f"point mode {point_mode} not implemented")
        pts.append(np.array(p))
    return np.stack(pts)


def quantise_image(query_image_1024):
    """ProtoSAM.py:651-660: [1,3,1024,1024] fp32 -> uint8 HWC via per-image min-max (float32 arithmetic)."""
    q = query_image_1024[0].permute(1, 2, 0).cpu().numpy()
    return ((q - q.min()) / (q.max() - q.min()) * 255).astype(np.uint8)


def sam_preprocess(img_u8_hwc):
    """predictor.py:56-58,88 + sam.py:163-173: normalise, zero-pad bottom / right to 1024 (input already at long side 1024)."""
    x = torch.as_tensor(img_u8_hwc).permute(2, 0, 1).contiguous()[None]
    mean = torch.tensor(PIXEL_MEAN).view(-1, 1, 1)
    std = torch.tensor(PIXEL_STD).view(-1, 1, 1)
    x = (x - mean) / std
    return F.pad(x, (0, 1024 - x.shape[-1], 0, 1024 - x.shape[-2]))


def apply_image(img_u8_hwc, target=1024):
    """utils/transforms.py:32-38: PIL bilinear resize to long side `target` (torchvision's resize(to_pil_image(.)))."""
    from PIL import Image
    h, w = img_u8_hwc.shape[:2]
    scale = target * 1.0 / max(h, w)
    nh, nw = int(h * scale + 0.5), int(w * scale + 0.5)
    if (nh, nw) == (h, w):
        return np.array(img_u8_hwc)
    return np.array(Image.fromarray(img_u8_hwc).resize((nw, nh), Image.BILINEAR))


def mask_prompts(cc):
    """ProtoSAM.get_sam_input_mask (:452-466) + the per-mask preparation of predict_w_masks (:471-479):
    cv2.resize(mask, (256, 256), INTER_NEAREST) samples source pixel floor(dst * 4) (cv2 is absent: restated), then
    10 / -8 are written into the float array and the array is cast with `.astype(np.uint8)`."""
    out = []
    for cc_id in np.unique(cc[1]):
        if cc_id == 0:
            continue
        m = (cc[1] == cc_id).astype(np.float32)[::4, ::4].copy()
        m[m == 1] = 10
        m[m == 0] = -8
        with np.errstate(invalid="ignore"):
            out.append(m[None, ...].astype(np.uint8))
    return out


def protosam_forward(query_image, output_logits, sam_sd, sam_type="vit_b", use_bbox=True, use_points=True,
                     point_mode="both", use_cca=False, postprocess="upstream", encoder_depth=None, taps=None,
                     features=None, use_mask=False, use_neg_points=False, num_points=1):
    """ProtoSAM.forward (models/ProtoSAM.py:536-678) after the coarse model: `output_logits` [1,2,H,W] is what
    `self.coarse_segmentation_model(input)` returned. Returns (pred [H,W] float {0,1}, scores list)."""
    original_size = query_image.shape[-2]
    if tuple(query_image.shape[-2:]) != (1024, 1024):                                   # :592-594
        query_image = F.interpolate(query_image, size=(1024, 1024), mode="bilinear")
        output_logits = F.interpolate(output_logits, size=(1024, 1024), mode="bilinear")
    output_p = output_logits.softmax(dim=1)                                               # :599-602
    _pred = output_p.argmax(dim=1)[0].numpy()
    if use_cca:
        cc, conf = cca(_pred, output_logits), None
    else:
        cc, conf = get_connected_components(_pred, output_logits)
    if taps is not None:
        taps.update(output_p=output_p, coarse_pred=_pred, cc=cc, conf=conf)
    if _pred.max() == 0:                                                                  # :612-613
        return output_p.argmax(dim=1)[0], [0]
    bboxes = bbox_per_cc(cc) if use_bbox else [None] * cc[0]
    points = sam_input_points(cc, output_p, point_mode, num_points) if use_points else [None] * cc[0]
    img_u8 = quantise_image(query_image)                                                   # :651-660 (sam_trans = identity)
    if features is None:
        features = oenc.image_encoder(sam_preprocess(img_u8), sam_sd, model_type=sam_type, depth=encoder_depth)
    if taps is not None:
        taps.update(bboxes=bboxes, points=points, img_u8=img_u8, features=features, low_res=[], masks=[])
    masks, scores = [], []
    if use_mask and not (use_points or use_bbox):                                          # :468-498, 664-665
        for in_mask in mask_prompts(cc):
            m, sc, low = odec.predict(sam_sd, features, None, None, None, multimask_output=True,
                                      original_size=(1024, 1024), variant=postprocess, mask_input=in_mask)
            k = int(sc.argmax())
            masks.append(m[k].numpy())
            scores.append(sc[k].item())
            if taps is not None:
                taps["low_res"].append(low)
                taps["masks"].append(m[k].numpy())
        points, bboxes = [], []
    negs = sam_neg_points(cc, output_p) if (use_neg_points and use_points) else [None] * len(points)
    if taps is not None:
        taps["neg_points"] = negs
    for point, box, neg in zip(points, bboxes, negs):                                      # :505-527
        labels = np.array([1] * len(point)) if point is not None else None
        if use_neg_points:                                                                 # :508-511
            neg_pts = [npt for npt in neg if None not in npt]
            point = np.vstack([point, *neg_pts])
            labels = np.array([1] * (len(point) - len(neg_pts)) + [0] * len(neg_pts))
        m, s, low = odec.predict(sam_sd, features, point, labels, box, multimask_output=not use_cca,
                                 original_size=(1024, 1024), variant=postprocess)
        masks.append(m[0].numpy())                                                         # best_pred_idx = 0
        scores.append(s[0].item())
        if taps is not None:
            taps["low_res"].append(low)
            taps["masks"].append(m[0].numpy())
    pred = torch.tensor(sum(masks) > 0).float()                                            # :669-672
    pred = F.interpolate(pred[None, None], size=original_size, mode="nearest")[0][0]       # :676
    return pred, scores


def protomedsam_forward(query_image, output_logits, sam_sd, sam_type="vit_b", use_cca=True, encoder_depth=None, taps=None):
    """ProtoMedSAM.forward after the coarse model (models/ProtoMedSAM.py:122-222; PARITY UNPINNED as a whole -- the file
    imports the absent pip `segment_anything` and cv2 -- but every stage it calls is pinned elsewhere)."""
    original_size = query_image.shape[-2]
    if tuple(query_image.shape[-2:]) != (1024, 1024):
        query_image = F.interpolate(query_image, size=(1024, 1024), mode="bilinear")
        output_logits = F.interpolate(output_logits, size=(1024, 1024), mode="bilinear")
    output_p = output_logits.softmax(dim=1)                       # need_softmax(ALP logits) is True (:178-179)
    _pred = output_p.argmax(dim=1)[0].numpy()
    cc = cca(_pred, output_p) if use_cca else get_connected_components(_pred, output_p)[0]   # softmax applied AGAIN inside
    if _pred.max() == 0:
        return F.interpolate(output_p, size=original_size, mode="bilinear").argmax(dim=1)[0], [0]
    H, W = query_image.shape[-2:]
    bbox = bbox_per_cc(cc) / np.array([W, H, W, H]) * 1024
    qi = (query_image - query_image.min()) / (query_image.max() - query_image.min())
    feats = oenc.image_encoder(qi, sam_sd, model_type=sam_type, depth=encoder_depth)
    box = torch.as_tensor(bbox, dtype=torch.float)[:, None, :]
    sparse, dense = odec.prompt_encoder(sam_sd, None, box)
    low, conf = odec.mask_decoder(sam_sd, feats, odec.dense_pe(sam_sd), sparse, dense, False)
    pr = F.interpolate(torch.sigmoid(low), size=(H, W), mode="bilinear", align_corners=False).squeeze()
    seg = torch.tensor((pr.numpy() > 0.5).astype(np.uint8))
    if taps is not None:
        taps.update(low=low, cc=cc)
    seg = F.interpolate(seg[None, None], size=original_size, mode="nearest")[0][0]
    return seg, [conf.numpy()]
