"""Host-side helpers of the automatic mask generator (names of models/segment_anything/utils/amg.py).

Only the pieces that remain on the host once the per-candidate reductions run on the GPU (`psam_mask_stats`): the point
grids (:179-199), crop boxes of layer 0 (:202-237), XYXY->XYWH (:92-96), the uncompressed column-major RLE
(:108-153) and box NMS. `torchvision.ops.batched_nms` (a dependency absent from /root/reference; all categories are 0 in
the generator, so it is plain NMS) is restated from its published contract: visit boxes by decreasing score (stable),
drop every later box whose IoU with a kept box exceeds the threshold, areas = (x2-x1)*(y2-y1) in fp32.
"""
import numpy as np


def build_point_grid(n_per_side):
    """n x n points at the cell centres of the unit square, x fastest. [n*n, 2] float64 (x, y)."""
    half = 1 / (2 * n_per_side)
    side = np.linspace(half, 1 - half, n_per_side)
    xs, ys = np.meshgrid(side, side)
    return np.stack([xs, ys], axis=-1).reshape(-1, 2)


def build_all_layer_point_grids(n_per_side, n_layers, scale_per_layer):
    return [build_point_grid(int(n_per_side / (scale_per_layer ** i))) for i in range(n_layers + 1)]


def box_xyxy_to_xywh(box):
    b = np.array(box, copy=True)
    b[2] -= b[0]
    b[3] -= b[1]
    return b


def batch_iterator(batch_size, *args):
    n = len(args[0])
    assert all(len(a) == n for a in args), "Batched iteration must have inputs of all the same size."
    for lo in range(0, n, batch_size):
        yield [a[lo:lo + batch_size] for a in args]


def mask_to_rle(mask):
    """bool [H, W] -> {"size": [H, W], "counts": [...]}: run lengths in column-major (Fortran) order, starting with
    the number of leading zeros (0 if the first pixel is set) - the pycocotools uncompressed format."""
    h, w = mask.shape
    flat = np.asarray(mask, dtype=bool).T.reshape(-1)
    change = np.flatnonzero(flat[1:] != flat[:-1]) + 1
    edges = np.concatenate([[0], change, [h * w]])
    counts = np.diff(edges).tolist()
    if flat.size and flat[0]:
        counts = [0] + counts
    return {"size": [h, w], "counts": counts}


def rle_to_mask(rle):
    h, w = rle["size"]
    counts = np.asarray(rle["counts"], dtype=np.int64)
    vals = (np.arange(len(counts)) & 1).astype(bool)
    return np.repeat(vals, counts).reshape(w, h).T


def area_from_rle(rle):
    return sum(rle["counts"][1::2])


def nms_xyxy(boxes, scores, iou_threshold):
    """-> indices kept, by decreasing score. boxes [n,4] (x0,y0,x1,y1), scores [n]."""
    b = np.asarray(boxes, dtype=np.float32).reshape(-1, 4)
    order = np.argsort(-np.asarray(scores, dtype=np.float32), kind="stable")
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    dead = np.zeros(len(b), bool)
    keep = []
    for pos, i in enumerate(order):
        if dead[i]:
            continue
        keep.append(int(i))
        rest = order[pos + 1:]
        rest = rest[~dead[rest]]
        if rest.size == 0:
            continue
        iw = np.maximum(np.float32(0), np.minimum(b[i, 2], b[rest, 2]) - np.maximum(b[i, 0], b[rest, 0]))
        ih = np.maximum(np.float32(0), np.minimum(b[i, 3], b[rest, 3]) - np.maximum(b[i, 1], b[rest, 1]))
        inter = iw * ih
        with np.errstate(divide="ignore", invalid="ignore"):
            iou = inter / (area[i] + area[rest] - inter)
        dead[rest[iou.astype(np.float64) > iou_threshold]] = True
    return np.asarray(keep, dtype=np.int64)


def generate_crop_boxes(im_size, n_layers, overlap_ratio):
    """utils/amg.py:202-237: the image box, then (2**i)**2 overlapping crops for layer i. -> (boxes XYXY, layer per box)."""
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
        xs = [int((cw - overlap) * i) for i in range(n)]
        ys = [int((ch - overlap) * i) for i in range(n)]
        for x0, y0 in product(xs, ys):
            boxes.append([x0, y0, min(x0 + cw, im_w), min(y0 + ch, im_h)])
            layers.append(i_layer + 1)
    return boxes, layers


def is_box_near_crop_edge(boxes, crop_box, orig_box, atol=20.0):
    """utils/amg.py:78-88: a box edge within `atol` of the crop's edge but not of the image's. boxes int [n,4] in the crop's
    frame -> bool [n]. (torch.isclose with rtol = 0 is |a - b| <= atol.)"""
    b = np.asarray(boxes, dtype=np.float32).reshape(-1, 4) + np.array([[crop_box[0], crop_box[1], crop_box[0], crop_box[1]]],
                                                                      dtype=np.float32)
    near_crop = np.abs(b - np.asarray(crop_box, dtype=np.float32)[None, :]) <= atol
    near_img = np.abs(b - np.asarray(orig_box, dtype=np.float32)[None, :]) <= atol
    return np.any(near_crop & ~near_img, axis=1)
