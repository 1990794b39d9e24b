"""`ProtoSAM` pipeline on the HIP path, with the reference's class API.

Mirrors /root/reference/models/ProtoSAM.py: `ALPNetInput` (:59-79), `ALPNetOutput` (:81-92), `InputFactory` (:112-130),
`ModelWrapper` / `ALPNetWrapper` (:133-168) and `ProtoSAM` (:184-678) with the same constructor arguments, the same
`forward(query_image, coarse_model_input, degrees_rotate=0) -> (pred [H,W] float {0,1} on device, scores list)`
contract, the same ValueError / AssertionError conditions.

What changes is where the work happens. The reference leaves the GPU four times per slice (argmax -> numpy -> cv2
connected components -> per-component python loops -> uint8 image -> `predictor.set_image` -> one `predict` per
component -> numpy -> tensor; ProtoSAM.py:602-676). Here every stage stays on the device:

  coarse logits --prob_argmax--> output_p, pred(u8) --ccl--> labels + per-component table (area, bbox, centroid sums,
  confidence, most-confident pixel) --async D2H of the table only-->           (host: number of components, prompts)
  query --bilinear 1024--> minmax --quantise+normalise+im2col--> SAM image encoder        (enqueued BEFORE the host waits
  for the table, so the wait overlaps the encoder)  --> prompt tokens --> batched two-way decoder over all components
  --> fused upsample / threshold / union / nearest-resize --> pred.

Mask prompts (`use_mask=True` with points and boxes off, ProtoSAM.py:468-498,664-665): each component's mask, nearest-
sampled to 256x256, goes through `psam_mask_downscale` and the decoder without sparse prompts; the best-scoring of the
three masks is kept per component. With points or boxes on, the reference overwrites the mask-prompt result (:667-668),
so `use_mask` then changes nothing and its work is skipped.
Negative points (`use_neg_points=True`, ProtoSAM.py:361-372,395-419,508-511): `psam_neg_points` finds, on the device, the
most confident background pixel of each component's 10-pixel dilation ring and the global one (p_bg >= 0.95); components
whose prompt sets end up with different token counts are decoded in separate batches.
`degrees_rotate != 0` runs the coarse model on the rotated query and rotates its logits back (protosam_amd/rotate.py).
`num_points_for_sam` = k > 1: the k most confident pixels of every component (`_topk_points`: the reference's own host-side
`torch.topk` call on the kernels' probabilities), k = 1 comes from the component table on the device.
Unsupported (outside SURVEY §8's hot path, raise NotImplementedError): `debug` plotting, training mode.
"""
import os
from abc import ABC, abstractmethod

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .grid_proto_fewshot import FewShotSeg
from .rotate import reverse_tensor, rotate_tensor_no_crop
from .sam_wrapper import SamWrapper
from .segment_anything import SamPredictor, sam_model_registry
from .segment_anything.utils.transforms import ResizeLongestSide

CONF_MODE = "conf"
CENTROID_MODE = "centroid"
BOTH_MODE = "both"
POINT_MODES = (CONF_MODE, CENTROID_MODE, BOTH_MODE)

TYPE_ALPNET = "alpnet"
TYPE_SAM = "sam"

MAX_COMPONENTS = 256          # fast path: table rows per slice that travel D2H every step
MAX_COMPONENTS_LARGE = 4096   # psam_ccl's limit; used for a slice that overflows the fast table
DECODER_CHUNK = 256           # prompt sets per decoder call (workspace ~ 25 MB per prompt set)
MAX_NEG_COMPONENTS = 64   # psam_neg_points launches one tile grid per component


class StageTimer:
    """Optional per-stage GPU timing of `forward_batch` (bench.py): HIP events on the launch stream at the stage boundaries;
    the time between two consecutive marks is booked on the later one. Off (None) by default: no events are recorded."""

    def __init__(self):
        self.marks = []

    def mark(self, name):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.marks.append((name, e))

    def summary(self):
        """{stage: total ms} -- call after a device synchronize."""
        out = {}
        for (_, e0), (name, e1) in zip(self.marks[:-1], self.marks[1:]):
            if name != "start":
                out[name] = out.get(name, 0.0) + e0.elapsed_time(e1)
        return out


STAGE_TIMER = None


def _mark(name):
    if STAGE_TIMER is not None:
        STAGE_TIMER.mark(name)


class SegmentationInput(ABC):
    @abstractmethod
    def set_query_images(self, query_images):
        pass

    def to(self, device):
        pass


class SegmentationOutput(ABC):
    @abstractmethod
    def get_prediction(self):
        pass


class ALPNetInput(SegmentationInput):
    def __init__(self, support_images, support_labels, query_images, isval, val_wsize, show_viz=False, supp_fts=None):
        self.supp_imgs = [support_images]
        self.fore_mask = [support_labels]
        self.back_mask = [[1 - sup_labels for sup_labels in support_labels]]
        self.qry_imgs = [query_images]
        self.isval = isval
        self.val_wsize = val_wsize
        self.show_viz = show_viz
        self.supp_fts = supp_fts

    def set_query_images(self, query_images):
        self.qry_imgs = [query_images]

    def to(self, device):
        self.supp_imgs = [[supp_img.to(device) for way in self.supp_imgs for supp_img in way]]
        self.fore_mask = [[fore_mask.to(device) for way in self.fore_mask for fore_mask in way]]
        self.back_mask = [[back_mask.to(device) for way in self.back_mask for back_mask in way]]
        self.qry_imgs = [qry_img.to(device) for qry_img in self.qry_imgs]
        if self.supp_fts is not None:
            self.supp_fts = self.supp_fts.to(device)


class ALPNetOutput(SegmentationOutput):
    def __init__(self, pred, align_loss, sim_maps, assign_maps, proto_grid, supp_fts, qry_fts):
        self.pred = pred
        self.align_loss = align_loss
        self.sim_maps = sim_maps
        self.assign_maps = assign_maps
        self.proto_grid = proto_grid
        self.supp_fts = supp_fts
        self.qry_fts = qry_fts

    def get_prediction(self):
        return self.pred


class SAMWrapperInput(SegmentationInput):
    """ProtoSAM.py:94-109."""

    def __init__(self, image, image_labels):
        self.image = image
        self.image_labels = image_labels

    def set_query_images(self, query_images):
        B, C, H, W = query_images.shape
        if isinstance(query_images, torch.Tensor):
            query_images = query_images.cpu().detach().numpy()
        assert B == 1, "batch size must be 1"
        query_images = (query_images - query_images.min()) / (query_images.max() - query_images.min()) * 255
        self.image = np.transpose(query_images.astype(np.uint8)[0], (1, 2, 0))

    def to(self, device):
        pass


class InputFactory(ABC):
    @staticmethod
    def create_input(input_type, query_image, support_images=None, support_labels=None, isval=False, val_wsize=None,
                     show_viz=False, supp_fts=None, original_sz=None, img_sz=None, gts=None):
        if input_type == TYPE_ALPNET:
            return ALPNetInput(support_images, support_labels, query_image, isval, val_wsize, show_viz, supp_fts)
        elif input_type == TYPE_SAM:                                            # ProtoSAM.py:118-128
            qimg = query_image.detach().cpu().numpy().copy()
            B, C, H, W = qimg.shape
            assert B == 1, "batch size must be 1"
            gts = gts.detach().cpu().numpy().astype(np.uint8).reshape(H, W)
            assert np.unique(gts).shape[0] <= 2, "support labels must be binary"
            gts[gts > 0] = 1
            qimg = qimg.reshape(H, W, C)   # sic: a reshape, not a transpose (ProtoSAM.py:126)
            qimg = (qimg - qimg.min()) / (qimg.max() - qimg.min()) * 255
            return SAMWrapperInput(qimg.astype(np.uint8), gts)
        else:
            raise ValueError("input_type not supported")


class ModelWrapper(ABC):
    def __init__(self, model):
        self.model = model

    def __call__(self, input_data):
        pass

    def state_dict(self):
        return self.model.state_dict()

    def load_state_dict(self, state_dict):
        self.model.load_state_dict(state_dict)

    def eval(self):
        self.model.eval()

    def train(self):
        self.model.train()

    def parameters(self):
        pass


class ALPNetWrapper(ModelWrapper):
    def __init__(self, model: FewShotSeg):
        super().__init__(model)

    def __call__(self, input_data: ALPNetInput):
        output = self.model(**input_data.__dict__)
        output = ALPNetOutput(*output)
        return output.pred

    def parameters(self):
        return self.model.encoder.parameters()

    def train(self):
        self.model.encoder.train()


class SamWrapperWrapper(ModelWrapper):
    """ProtoSAM.py:170-182: makes SamWrapper's best mask look like 2-class logits [1, 2, H, W]. The mask never leaves
    the device here (the reference builds the tensor on the host, :176)."""

    def __init__(self, model: SamWrapper):
        super().__init__(model)

    def __call__(self, input_data: SAMWrapperInput):
        pred = self.model(**input_data.__dict__, return_device=True).float()[None, None, ...]
        return torch.cat([1 - pred, pred], dim=1)

    def to(self, device):
        self.model.sam.to(device)


class ProtoSAM(nn.Module):
    def __init__(self, image_size, coarse_segmentation_model: ModelWrapper,
                 sam_pretrained_path="pretrained_model/sam_default.pth", num_points_for_sam=1, use_points=True,
                 use_bbox=False, use_mask=False, debug=False, use_cca=False, point_mode=CONF_MODE, use_sam_trans=True,
                 coarse_pred_only=False, alpnet_image_size=None, use_neg_points=False):
        super().__init__()
        if isinstance(image_size, int):
            image_size = (image_size, image_size)
        self.image_size = image_size
        self.coarse_segmentation_model = coarse_segmentation_model
        self.get_sam(sam_pretrained_path, use_sam_trans)
        self.num_points_for_sam = num_points_for_sam
        self.use_points = use_points
        self.use_bbox = use_bbox
        self.use_mask = use_mask
        self.use_neg_points = use_neg_points
        assert self.use_bbox or self.use_points or self.use_mask, "must use at least one of bbox, points, or mask"
        self.use_cca = use_cca
        self.point_mode = point_mode
        if self.point_mode not in POINT_MODES:
            raise ValueError(f"point mode must be one of {POINT_MODES}")
        self.debug = debug
        self.coarse_pred_only = coarse_pred_only
        if debug:
            raise NotImplementedError("debug plotting is outside the hot path")
        self.num_points_for_sam = int(num_points_for_sam)
        if self.num_points_for_sam < 1:
            raise ValueError("num_points_for_sam must be >= 1")
        self._mask_only = self.use_mask and not (self.use_points or self.use_bbox)
        # predict_w_masks writes 10 / -8 into a float array and hands it over `.astype(np.uint8)` (ProtoSAM.py:473-479):
        # whatever this platform's numpy makes of -8.0 (248 on x86-64) is what SAM sees
        with np.errstate(invalid="ignore"):
            self._mask_vals = tuple(float(v) for v in np.array([10.0, -8.0], dtype=np.float32).astype(np.uint8))
        if tuple(self.image_size) != (1024, 1024):
            raise NotImplementedError("image_size must be (1024, 1024) as in validation_protosam.py:220")
        self._ccl = None
        self._bufs = {}
        self.last_stats = {}
        # The SAM image encoder does not depend on the coarse model: on a second HIP stream it shares the GPU with DINOv2 + ALP +
        # connected components, whose small launches and part-filled rounds then cost nothing. Round 2 measured no gain (107.0 / 110.3
        # vs 108.2 / 107.8 slices/s); with round 3's kernels it is 150.6 / 151.1 vs 148.3 / 148.5 (same box, interleaved). The price:
        # the encoder runs on EVERY slice of the batch, before anybody knows which coarse masks are empty (the sequential order skips
        # those, ProtoSAM.py:612-613: on the sparse test volume that is 195 vs 163 slices/s). PSAM_OVERLAP_STREAMS: "auto" (default)
        # overlaps a call when the last four calls of this model had no empty slice (and no per-kernel timer is attached: concurrent
        # kernels blur those), "1" always, "0" never. Same results. One slice per call (the reference's convention) gains most:
        # 88.0 -> 100.6 slices/s, the coarse model's ~150 small launches run beside the encoder's GEMMs.
        self.overlap_streams = os.environ.get("PSAM_OVERLAP_STREAMS", "auto")
        self._dense_run = 0

    def get_sam(self, checkpoint_path, use_sam_trans):
        """ProtoSAM.py:205-220. `random:<vit_b|vit_l|vit_h>[:seed[:depth]]` builds seeded synthetic weights instead of reading a
        checkpoint (none exist offline); `random-heavy:...` the same with the heavy-tailed residual stream of synth.heavy_tail_sam_."""
        model_type = "vit_b"
        if checkpoint_path is not None and "vit_h" in checkpoint_path:
            model_type = "vit_h"
        if checkpoint_path is not None and checkpoint_path.startswith(("random:", "random-heavy:")):
            from .synth import heavy_tail_sam_, synth_state_dict
            parts = checkpoint_path.split(":")
            model_type = parts[1]
            seed = int(parts[2]) if len(parts) > 2 else 1234
            depth = int(parts[3]) if len(parts) > 3 else None   # test hook: truncated block stack
            self.sam = sam_model_registry[model_type](encoder_depth=depth)
            sd = synth_state_dict(self.sam, seed)
            if parts[0] == "random-heavy":
                heavy_tail_sam_(sd, seed)
            self.sam.load_state_dict(sd)
            self.sam.eval()
        else:
            self.sam = sam_model_registry[model_type](checkpoint=checkpoint_path).eval()
        self.predictor = SamPredictor(self.sam)
        self.sam.requires_grad_(False)
        if use_sam_trans:
            sam_trans = ResizeLongestSide(self.sam.image_encoder.img_size)
            sam_trans.pixel_mean = torch.tensor([0, 0, 0]).view(3, 1, 1)
            sam_trans.pixel_std = torch.tensor([1, 1, 1]).view(3, 1, 1)
        else:
            sam_trans = None
        self.sam_trans = sam_trans

    # ---- the reference's helper methods by name (ProtoSAM.py:222-289,349-533) ----------------------------------------------
    # `forward` does their work inside fused kernels (component table, psam_neg_points, one batched decoder call); these are for
    # callers that use the helpers directly, with the reference's arguments and return types. `conn_components` is the cv2-style
    # tuple of protosam_amd.utils.cca / get_connected_components.
    def get_bbox(self, pred):
        """ProtoSAM.py:222-240: the four corners [[y,x] x 4] of the non-zero pixels of pred [H,W]."""
        if isinstance(pred, np.ndarray):
            pred = torch.tensor(pred)
        indices = torch.nonzero(pred)
        min_x, max_x = indices[:, 1].min(), indices[:, 1].max()
        min_y, max_y = indices[:, 0].min(), indices[:, 0].max()
        return [[min_y, min_x], [min_y, max_x], [max_y, max_x], [max_y, min_x]]

    def get_bbox_per_cc(self, conn_components):
        """ProtoSAM.py:242-264: XYXY [min_x, min_y, max_x, max_y] per label 1 ... n-1."""
        bboxes = []
        labels = np.asarray(conn_components[1])
        for i in range(1, conn_components[0]):
            ys, xs = np.nonzero(labels == i)
            bboxes.append([xs.min(), ys.min(), xs.max(), ys.max()])
        return np.array(bboxes)

    def get_most_conf_points(self, output_p_fg, pred, k):
        """ProtoSAM.py:266-289: the k most confident pixels of output_p_fg [H,W] inside pred [H,W] -> (xy locations numpy [k,2],
        confidences list). `torch.topk` on the host, as the reference: the same order among equal probabilities."""
        output_p_fg, pred = output_p_fg.detach().cpu(), torch.as_tensor(pred).cpu()
        mask = pred.bool()
        masked = output_p_fg[mask]
        if masked.numel() == 0:
            return None, None
        confidences, indices = torch.topk(masked, k)
        locations = torch.nonzero(mask)[indices][:, [1, 0]]
        return locations.numpy(), [float(c.item()) for c in confidences]

    def get_sam_input_points(self, conn_components, output_p, get_neg_points=False, l=1):
        """ProtoSAM.py:349-450: per component the positive points of `point_mode` and, with `get_neg_points`, the most confident
        background pixel of its 10-pixel dilation ring (cv2.dilate with a 3 x 3 kernel, 10 iterations = one 21 x 21 box maximum,
        outside pixels ignored) followed by the global one (p_bg >= 0.95).
        -> (points [n_cc, n_pts, 2], labels, neg points list or None, neg labels or None)."""
        sam_input_points, sam_neg_points = [], []
        fg_p = output_p[0, 1].detach().cpu()
        labels_img = np.asarray(conn_components[1])
        if get_neg_points:
            bg_p = output_p[0, 0].detach().cpu().clone()
            bg_p[bg_p < 0.95] = 0
            glob_neg_points, _ = self.get_most_conf_points(bg_p, torch.where(bg_p > 0, 1, 0), 1)
        for cc_id in np.unique(labels_img):
            if cc_id == 0:
                continue
            pred = torch.tensor(labels_img == cc_id).float()
            if self.point_mode == CONF_MODE:
                points, confidences = self.get_most_conf_points(fg_p, pred, self.num_points_for_sam)
            elif self.point_mode == CENTROID_MODE:
                points = conn_components[3][cc_id][None, :]
            elif self.point_mode == BOTH_MODE:
                points, confidences = self.get_most_conf_points(fg_p, pred, self.num_points_for_sam)
                points = np.vstack([points, conn_components[3][cc_id][None, :]])
            else:
                raise NotImplementedError(f"point mode {self.point_mode} not implemented")
            sam_input_points.append(np.array(points))
            if get_neg_points:
                dil = torch.nn.functional.max_pool2d(pred[None, None], kernel_size=21, stride=1, padding=10)[0, 0]
                boundary = dil - pred                                                    # the ring outside the component
                neg_points, _ = self.get_most_conf_points(output_p[0, 0].detach().cpu(), boundary, l)
                if neg_points is not None and glob_neg_points is not None:
                    neg_points = np.vstack([neg_points, glob_neg_points])
                else:
                    neg_points = glob_neg_points if neg_points is None else neg_points
                sam_neg_points.append(neg_points)
            else:
                sam_neg_points = [None for _ in range(len(sam_input_points))]
        sam_input_labels = np.array([i + 1 for i, cc_points in enumerate(sam_input_points) for _ in range(len(cc_points))])
        sam_input_points = np.stack(sam_input_points)
        return sam_input_points, sam_input_labels, sam_neg_points, np.array([0] * len(sam_neg_points))

    def get_sam_input_mask(self, conn_components):
        """ProtoSAM.py:452-466: one {0,1} float mask per label >= 1 and the labels."""
        labels_img = np.asarray(conn_components[1])
        ids = [c for c in np.unique(labels_img) if c != 0]
        return np.stack([(labels_img == c).astype(np.float32) for c in ids]), np.array(ids)

    def _set_query_image(self, qry_img):
        assert qry_img.max() <= 255 and qry_img.min() >= 0 and qry_img.dtype == np.uint8
        self.predictor.set_image(qry_img)

    def predict_w_masks(self, sam_input_masks, qry_img, original_size):
        """ProtoSAM.py:468-498: every component's mask, nearest-sampled to 256 x 256 with the values {10, uint8(-8)}, as a mask prompt;
        the best-scoring of the three masks per component. (The reference encodes the same image once per mask; once is enough.)"""
        masks, scores = [], []
        self._set_query_image(qry_img)
        fg, bg = self._mask_vals
        for in_mask in sam_input_masks:
            in_mask = np.asarray(in_mask)
            H, W = in_mask.shape
            ys, xs = (np.arange(256) * H) // 256, (np.arange(256) * W) // 256        # cv2.resize(..., INTER_NEAREST)
            m = in_mask[ys][:, xs]
            prompt = np.where(m == 1, fg, bg).astype(np.uint8)
            mask, score, _ = self.predictor.predict(mask_input=prompt[None, ...], multimask_output=True)
            best = score.argmax()
            masks.append(mask[best])
            scores.append(score[best])
        return masks, scores

    def predict_w_points_bbox(self, sam_input_points, bboxes, sam_neg_input_points, qry_img, pred, return_logits=False):
        """ProtoSAM.py:500-533: one `predictor.predict` per component with its points (+ negative points) and box; mask 0 of the
        result is kept (three masks unless `use_cca`). qry_img: uint8 HWC. `last_stats` keeps the low-res logits and scores of the
        kept masks (`low_res` [n,256,256], `iou` [n])."""
        masks, scores, lows = [], [], []
        self._set_query_image(qry_img)
        for point, bbox_xyxy, neg_point in zip(sam_input_points, bboxes, sam_neg_input_points):
            points = point
            point_labels = np.array([1] * len(point)) if point is not None else None
            if self.use_neg_points:
                neg_points = [npoint for npoint in neg_point if None not in npoint]
                points = np.vstack([point, *neg_points])
                point_labels = np.array([1] * len(point) + [0] * len(neg_points))
            mask, score, low = self.predictor.predict(point_coords=points, point_labels=point_labels,
                                                      box=bbox_xyxy if bbox_xyxy is not None else None,
                                                      return_logits=return_logits, multimask_output=False if self.use_cca else True)
            masks.append(mask[0])
            scores.append(score[0])
            lows.append(low[0])
        self.last_stats = dict(low_res=np.stack(lows) if lows else None, iou=np.array(scores), n_prompts=len(masks))
        return masks, scores

    # ---- host-side prompt assembly from the component table -----------------------------------------------------------
    def _topk_points(self, pfg, labels, S, ids, k):
        """num_points_for_sam = k > 1 (ProtoSAM.get_most_conf_points, ProtoSAM.py:266-289): the k most confident pixels of each
        component in `ids` (labels of csrc/ccl.hip). The reference takes `torch.topk(output_p_fg[mask], k)` on the host; so does
        this - the same call on the same raster-ordered values, hence the same order among equal probabilities (a saturated
        soft-max has many) - on the probabilities and labels the kernels produced. Not the default (k = 1 comes out of the
        component table on the device); two 4 MB copies per slice."""
        pf = pfg.reshape(S, S).cpu()
        lab = labels.view(S, S).cpu()
        out = {}
        for cid in ids:
            mask = lab == cid
            conf, idx = torch.topk(pf[mask], k)            # (raises for a component of fewer than k pixels, as the reference)
            out[cid] = torch.nonzero(mask)[idx][:, [1, 0]].numpy().astype(np.float64)
        return out

    def _prompts_from_table(self, tab, neg_keys=None, topk=None):
        """tab: fp64 numpy table of csrc/ccl.hip; neg_keys: int64 numpy keys of psam_neg_points (use_neg_points).
        Returns per kept component a list of (x, y) and a list of labels in the prompt kernel's convention (1 = positive
        point, 0 = negative point, 2/3 = box corners, -1 = padding point) following get_sam_input_points (:349-450),
        get_bbox_per_cc (:242-264), predict_w_points_bbox (:505-511) and PromptEncoder's padding rule; plus the rows."""
        n = int(tab[1])
        rows = tab[ops.CC_HDR:ops.CC_HDR + ops.CC_STRIDE * n].reshape(n, ops.CC_STRIDE)
        first = 0
        if self.use_cca:  # util/utils.py:496-541: keep the most confident component only
            first = int(tab[3])
            rows = rows[first:first + 1]
        glob = None
        if self.use_neg_points:
            if not self.use_points:
                raise TypeError("'NoneType' object is not iterable")   # ProtoSAM.py:509 iterates sam_neg_input_points[i] = None
            if len(neg_keys) < n + 1:
                raise RuntimeError("negative-point keys of fewer components than the table holds")
            glob = ops.decode_point_key(int(neg_keys[0]), 1024)
        coords, labels = [], []
        for k, r in enumerate(rows):
            c, lab = [], []
            if self.use_points:
                if self.point_mode in (CONF_MODE, BOTH_MODE):
                    if topk is not None:                    # component first + k of the table carries label first + k + 1
                        for pt in topk[first + k + 1]:
                            c.append([pt[0], pt[1]])
                            lab.append(1)
                    else:
                        c.append([r[8], r[9]])
                        lab.append(1)
                if self.point_mode in (CENTROID_MODE, BOTH_MODE):
                    c.append([r[1] / r[0], r[2] / r[0]])  # cv2 centroid: float64 mean of x, mean of y
                    lab.append(1)
                if self.use_neg_points:                     # ring point first, then the global one (:414-418)
                    ring = ops.decode_point_key(int(neg_keys[1 + first + k]), 1024)
                    if ring is None and glob is None:
                        raise TypeError("'NoneType' object is not iterable")   # neg_point is None at :509
                    for pt in (ring, glob):
                        if pt is not None:
                            c.append([float(pt[0]), float(pt[1])])
                            lab.append(0)
                if not self.use_bbox:
                    c.append([0.0, 0.0])
                    lab.append(-1)
            if self.use_bbox:
                c += [[r[3], r[4]], [r[5], r[6]]]
                lab += [2, 3]
            coords.append(c)
            labels.append(lab)
        # predictor.predict: apply_coords with original_size == 1024 is the identity; torch.as_tensor(dtype=float)
        return coords, labels, rows

    def _work_buffers(self, dev, B):
        key = (str(dev), B)
        if key not in self._bufs:
            self._bufs[key] = dict(
                fg_sum=torch.zeros(B, dtype=torch.int32, device=dev),
                prob=torch.empty((B, 2, 1024, 1024), dtype=torch.float32, device=dev),
                pred=torch.empty((B, 1024, 1024), dtype=torch.uint8, device=dev),
                q1024=torch.empty((B, 3, 1024, 1024), dtype=torch.float32, device=dev),
                mm=torch.empty(2 * B, dtype=torch.int32, device=dev),
                patches=torch.empty((B * 4096, 768), dtype=torch.float16, device=dev),
                fg_host=torch.zeros(B, dtype=torch.int32).pin_memory(), fg_event=torch.cuda.Event(),
                event=torch.cuda.Event(), sam_done=torch.cuda.Event())
        if self._ccl is None or self._ccl.slots < B:
            self._ccl = ops.CclWorkspace(1024, 1024, MAX_COMPONENTS, dev, slots=max(B, 1))
        return self._bufs[key]

    def _ccl_overflow(self, b, bufs, output_p, pred, S):
        """Slice `b` has more than MAX_COMPONENTS connected components: label it again with a MAX_COMPONENTS_LARGE table
        (synchronous, rare) and refresh the per-slice side products that were derived from the truncated labelling."""
        dev = pred.device
        if getattr(self, "_ccl_big", None) is None:
            self._ccl_big = ops.CclWorkspace(S, S, MAX_COMPONENTS_LARGE, dev, slots=1)
        big = self._ccl_big
        ops.ccl(pred[b], output_p[b, 1], big, fg_sum=bufs["fg_sum"][b:b + 1], slot=0)
        if self._mask_only:
            bufs["lab256"][b].copy_(big.labels.view(S, S)[::4, ::4])
        tab = big.tabs[0].cpu().numpy()
        if int(tab[0]) > int(tab[1]):
            raise RuntimeError(f"{int(tab[0])} connected components in one coarse mask exceed the table capacity "
                               f"{MAX_COMPONENTS_LARGE} (csrc/ccl.hip)")
        return tab

    def _side_stream(self, dev):
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=dev)
        return self._side

    def _sam_features(self, query_images, bufs, B, S):
        """resize -> min/max -> uint8 quantise -> SAM normalise -> im2col -> image encoder (ProtoSAM.py:592-593,651-660,
        predictor.set_image) on the current stream. -> token-major embeddings [B, 4096, 256]."""
        sam = self.sam
        q = query_images.float().contiguous()
        if tuple(q.shape[-2:]) != (S, S):
            q = ops.bilinear_nchw(q, S, S, out=bufs["q1024"][:B])
        mm, patches = bufs["mm"][:2 * B], bufs["patches"][:B * 4096]       # B may be a sub-batch (non-empty slices only)
        ops.minmax(q, B, mm=mm)
        enc = sam.image_encoder
        if getattr(enc, "split_fp16", False) or getattr(enc, "gemm_x3", False):
            # the quantised pixel values themselves (0 ... 255 are exact in fp16); Sam.preprocess' (x - mean) / std lives in the patch-
            # embedding weights (ImageEncoderViT._patch_raw): no fp16 rounding of the normalised pixels
            ops.sam_patchify(q, mm, S, enc.patch_size, (0.0, 0.0, 0.0), (1.0, 1.0, 1.0), True, out=patches)
            return enc.encode_patches(patches, B, raw_norm=(tuple(sam._mean_host), tuple(sam._std_host)))
        ops.sam_patchify(q, mm, S, enc.patch_size, sam._mean_host, sam._std_host, True, out=patches)
        return enc.encode_patches(patches, B)

    def forward(self, query_image, coarse_model_input, degrees_rotate=0):
        """Reference contract (ProtoSAM.py:536-678): one query slice [1,3,H,W] -> (pred [H,W] float {0,1}, scores)."""
        if self.training:
            raise NotImplementedError("training-mode outputs (logits) are outside the inference hot path")
        if self.coarse_pred_only:
            output_logits = self._coarse_logits(query_image, coarse_model_input, degrees_rotate)
            return self._coarse_only(output_logits, query_image.shape[-2])
        return self.forward_batch(query_image, coarse_model_input, degrees_rotate)[0]

    def _coarse_logits(self, query_images, coarse_model_input, degrees_rotate=0):
        """ProtoSAM.py:544-556: the coarse model sees the query rotated by `degrees_rotate` (expanded canvas resized back to
        H x W), its logits are rotated back and centre-cropped (protosam_amd/rotate.py); identity at 0 degrees."""
        if isinstance(coarse_model_input, (list, tuple)):
            # mixed-support batch (forward_batch): [(ALPNetInput, n), ...] in batch order - one encoder forward for all slices, the
            # prototype match per support set (FewShotSeg.forward_groups)
            if degrees_rotate != 0:
                raise NotImplementedError("rotation TTA with a mixed-support batch")
            alp = getattr(self.coarse_segmentation_model, "model", None)
            if not hasattr(alp, "forward_groups"):
                raise TypeError("a list of (input, count) pairs needs a coarse model with forward_groups (ALPNetWrapper(FewShotSeg))")
            return alp.forward_groups(query_images, [(i.supp_imgs, i.fore_mask, i.back_mask, i.isval, i.val_wsize, n)
                                                     for i, n in coarse_model_input])
        if degrees_rotate == 0:
            coarse_model_input.set_query_images(query_images)
            return self.coarse_segmentation_model(coarse_model_input)
        rotated, (rot_h, rot_w) = rotate_tensor_no_crop(query_images, degrees_rotate)
        coarse_model_input.set_query_images(rotated)
        logits_rot = self.coarse_segmentation_model(coarse_model_input)
        return reverse_tensor(logits_rot, rot_h, rot_w, -degrees_rotate)

    @torch.no_grad()
    def forward_batch(self, query_images, coarse_model_input, degrees_rotate=0):
        """MI355X extension: B independent query slices [B,3,H,W] go through every stage as one batch (the ViT GEMMs see
        M = B*tokens rows, the mask decoder sees all components of all slices at once). `coarse_model_input`: one input (all
        slices share its support set) or a list of (input, n) pairs in batch order (the first n slices belong to the first support
        set, ...: both encoders, the connected components, SAM and the decoder do not depend on the support; only the prototype
        match does). Slices never interact, so each result equals the per-slice `forward`. Returns a list of (pred, scores)."""
        B = query_images.shape[0]
        original_size = query_images.shape[-2]
        dev = query_images.device
        if self.coarse_pred_only:                                               # ProtoSAM.py:580-590: nothing of SAM runs
            return self._coarse_only_batch(self._coarse_logits(query_images, coarse_model_input, degrees_rotate), original_size)
        bufs = self._work_buffers(dev, B)
        sam = self.sam
        S = sam.image_encoder.img_size
        # 0. The SAM image encoder depends on the query image only, not on the coarse model: with `overlap_streams` it is
        #    enqueued on a second HIP stream so the two ViTs share the GPU (optional, see __init__)
        main = torch.cuda.current_stream(dev)
        mode = self.overlap_streams
        if isinstance(mode, str):
            mode = True if mode == "1" else False if mode == "0" else (self._dense_run >= 4 and STAGE_TIMER is None and not ops.TIMERS and ops.GEMM_TIMER is None)
        side = self._side_stream(dev) if mode else None
        feat_tok = None
        if side is not None:
            side.wait_stream(main)                                              # inputs / earlier work on `main` are visible
            with torch.cuda.stream(side):
                feat_tok = self._sam_features(query_images, bufs, B, S)
                bufs["sam_done"].record(side)
        _mark("start")
        output_logits = self._coarse_logits(query_images, coarse_model_input, degrees_rotate)   # [B,2,H,W]
        _mark("coarse: DINOv2 + ALP")
        # 1. (bilinear to 1024) -> softmax -> argmax                               ProtoSAM.py:592-602
        bufs["fg_sum"].zero_()
        output_p, pred = ops.prob_argmax(output_logits.float().contiguous(), S, S, prob=bufs["prob"], pred=bufs["pred"],
                                         fg_sum=bufs["fg_sum"])
        bufs["fg_host"].copy_(bufs["fg_sum"], non_blocking=True)                # foreground pixel count per slice
        bufs["fg_event"].record()
        # 2. connected components + per-component statistics; the tables go to pinned host memory asynchronously
        cw = self._ccl
        ops.ccl_batch(pred[:B], output_p[:B], cw, fg_sum=bufs["fg_sum"])        # one chain of six launches for the batch
        for b in range(B):
            if self.use_neg_points:
                nk = bufs.setdefault("neg_keys", torch.empty((B, MAX_NEG_COMPONENTS + 1), dtype=torch.int64, device=dev))
                ops.neg_points(cw, output_p[b, 0], cw.tabs[b], MAX_NEG_COMPONENTS, keys=nk[b], labels=cw.labels_b[b])
            if self._mask_only:   # cv2.resize(mask, (256, 256), INTER_NEAREST) samples pixel (4y, 4x)
                bufs.setdefault("lab256", torch.empty((B, S // 4, S // 4), dtype=torch.int32, device=dev))[b].copy_(
                    cw.labels_b[b].view(S, S)[::4, ::4])
        cw.tabs_host[:B].copy_(cw.tabs[:B], non_blocking=True)
        if self.use_neg_points:
            nkh = bufs.setdefault("neg_keys_host", torch.empty((B, MAX_NEG_COMPONENTS + 1), dtype=torch.int64).pin_memory())
            nkh.copy_(bufs["neg_keys"], non_blocking=True)
        bufs["event"].record()
        _mark("softmax / argmax + connected components")
        # 3./4. image hand-off + SAM image encoder (already running on the side stream, or enqueued here before the host looks
        #       at the component tables)
        #       A slice whose coarse mask is empty never reaches SAM (ProtoSAM.py:612-613 returns before `set_image`): only the
        #       non-empty slices are encoded. The foreground counts arrive while the CCL kernels above keep the GPU busy.
        feat_row = list(range(B))                                               # slice -> row of feat_tok
        if feat_tok is None:
            bufs["fg_event"].synchronize()
            keep = [b for b in range(B) if int(bufs["fg_host"][b]) > 0]
            if len(keep) == B:
                feat_tok = self._sam_features(query_images, bufs, B, S)
            elif keep:
                sub = query_images[torch.tensor(keep, device=dev)]
                feat_tok = self._sam_features(sub, bufs, len(keep), S)
                feat_row = [-1] * B
                for i, b in enumerate(keep):
                    feat_row[b] = i
        _mark("image hand-off + SAM image encoder")
        # 5. host: number of components and prompts per slice
        bufs["event"].synchronize()
        if side is not None:
            main.wait_event(bufs["sam_done"])                                   # the decoder below consumes feat_tok on `main`
            feat_tok.record_stream(main)
        tabs = cw.tabs_host[:B].numpy()
        # (what "auto" overlap looks at next time)
        self._dense_run = self._dense_run + 1 if all(int(tabs[b][0]) > 0 for b in range(B)) else 0
        results = [None] * B
        coords, labels, img_idx, slice_idx, spans = [], [], [], [], []   # img_idx: row of feat_tok, slice_idx: slice
        stats = []
        for b in range(B):
            tab = tabs[b]
            overflow = int(tab[0]) > int(tab[1])
            if overflow:
                # more components than the fast table holds (cv2 + the reference's per-component loop have no limit,
                # util/utils.py:474-494, ProtoSAM.py:505): redo this slice with the large table
                tab = self._ccl_overflow(b, bufs, output_p, pred, S)
            n_found, n = int(tab[0]), int(tab[1])
            stats.append(dict(n_components=n_found, fg_pixels=int(tab[2]), n_prompts=0))
            if n == 0:                                                          # ProtoSAM.py:612-613
                results[b] = (output_p[b].argmax(dim=0), [0])
                continue
            if self._mask_only:
                # component k of the table carries label k + 1 (csrc/ccl.hip); cca keeps the most confident one
                ids = [int(tab[3]) + 1] if self.use_cca else list(range(1, n + 1))
                spans.append((b, len(img_idx), len(ids)))
                labels += ids
                img_idx += [feat_row[b]] * len(ids)
                slice_idx += [b] * len(ids)
                stats[b].update(n_prompts=len(ids))
                continue
            neg_keys = None
            if self.use_neg_points:
                neg_keys = bufs["neg_keys_host"][b].numpy()
                if n > MAX_NEG_COMPONENTS or n_found > MAX_COMPONENTS:
                    # the fast path asked for the rings of the first MAX_NEG_COMPONENTS components of the fast table only (one
                    # tile grid per component): ask again for all of them, on the labelling the table in hand belongs to
                    # (the reference walks every component, ProtoSAM.py:395-419)
                    big = n_found > MAX_COMPONENTS
                    ws = self._ccl_big if big else cw
                    neg_keys = ops.neg_points(ws, output_p[b, 0], ws.tabs[0] if big else cw.tabs[b], n,
                                              labels=None if big else cw.labels_b[b]).cpu().numpy()
            topk = None
            if self.num_points_for_sam > 1 and self.use_points and self.point_mode in (CONF_MODE, BOTH_MODE):
                ids = [int(tab[3]) + 1] if self.use_cca else list(range(1, n + 1))
                topk = self._topk_points(output_p[b, 1], self._ccl_big.labels if overflow else cw.labels_b[b], S, ids,
                                         self.num_points_for_sam)
            c, l, rows = self._prompts_from_table(tab, neg_keys, topk)
            spans.append((b, len(img_idx), len(l)))
            coords += c
            labels += l
            img_idx += [feat_row[b]] * len(l)
            stats[b].update(n_prompts=len(l), table=rows, prompts=(c, l))
        self.last_stats = stats[0] if B == 1 else dict(per_slice=stats)
        if spans and self._mask_only:
            # 6m. mask prompts: dense embedding per component, no sparse prompts, best of the three masks
            #     (get_sam_input_mask :452-466, predict_w_masks :468-498)
            P = len(labels)
            pe = sam.prompt_encoder._packed()
            dpk = sam.mask_decoder._packed()
            iop = torch.tensor(img_idx, dtype=torch.int64).to(dev, non_blocking=True)
            ids = torch.tensor(labels, dtype=torch.int32).to(dev, non_blocking=True)
            fg, bg = self._mask_vals
            sop = torch.tensor(slice_idx, dtype=torch.int64).to(dev, non_blocking=True)
            masks = torch.empty((P, 4, 256, 256), dtype=torch.float32, device=dev)
            iou = torch.empty((P, 4), dtype=torch.float32, device=dev)
            zero_dense = torch.zeros(256, dtype=torch.float32, device=dev)
            for c0 in range(0, P, DECODER_CHUNK):
                # the dense prompt embedding and the decoder's image operand are [chunk, 4096, 256] fp32 (4 MiB per prompt set
                # each): built per chunk, so that memory stays bounded however many components a slice has
                c1 = min(c0 + DECODER_CHUNK, P)
                prompt = torch.where(bufs["lab256"][sop[c0:c1]] == ids[c0:c1, None, None], fg, bg).to(torch.float32)
                dense = sam.prompt_encoder.embed_masks_tokens(prompt[:, None])           # [chunk, 4096, 256]
                src = (feat_tok[iop[c0:c1]] + dense).contiguous()                         # mask_decoder.py:126-127
                del dense
                tokens = dpk["out_tok"].unsqueeze(0).expand(c1 - c0, -1, -1).contiguous()
                sam.mask_decoder.predict_masks_tokens(
                    src, pe["pe_tok"], tokens, zero_dense,
                    img_of_prompt=torch.arange(c1 - c0, dtype=torch.int32, device=dev), masks_out=masks[c0:c1],
                    iou_out=iou[c0:c1])
                del src
            best = iou[:, 1:].argmax(dim=1)                                               # score.argmax(), :494
            chosen = masks[torch.arange(P, device=dev), best + 1].unsqueeze(1).contiguous()   # [P,1,256,256]
            iou_host = iou[:, 1:].max(dim=1).values.cpu().numpy()
            for (b, start, cnt) in spans:
                out = ops.mask_union(chosen[start:start + cnt], 0, S, original_size, sam.variant_id(), sam.mask_threshold)
                results[b] = (out, [np.float32(v) for v in iou_host[start:start + cnt]])
            self.last_stats.update(low_res=masks, iou=iou, best=best, spans=spans)
        elif spans:
            P = len(labels)
            pe = sam.prompt_encoder._packed()
            dpk = sam.mask_decoder._packed()
            iop_all = torch.tensor(img_idx, dtype=torch.int32).to(dev, non_blocking=True)
            # 6. batched two-way decoder over all components of all slices (ProtoSAM.py:500-527); prompt sets of equal
            #    length share a batch (they differ only when some component has no ring / no global negative point)
            groups = {}
            for i, lab in enumerate(labels):
                groups.setdefault(len(lab), []).append(i)
            masks = iou = None
            for Ns, idx in groups.items():
                cg = np.asarray([coords[i] for i in idx], dtype=np.float64).astype(np.float32).reshape(len(idx), Ns, 2)
                lg = np.asarray([labels[i] for i in idx], dtype=np.int32).reshape(len(idx), Ns)
                tokens = ops.prompt_tokens(torch.from_numpy(cg).to(dev, non_blocking=True),
                                           torch.from_numpy(lg).to(dev, non_blocking=True), pe["G"], pe["type_emb"],
                                           dpk["out_tok"], len(idx), Ns, float(S))
                if len(groups) == 1 and P <= DECODER_CHUNK:
                    masks, iou, _ = sam.mask_decoder.predict_masks_tokens(feat_tok, pe["pe_tok"], tokens, pe["no_mask"],
                                                                          img_of_prompt=iop_all)
                else:
                    if masks is None:
                        masks = torch.empty((P, 4, 256, 256), dtype=torch.float32, device=dev)
                        iou = torch.empty((P, 4), dtype=torch.float32, device=dev)
                    it = torch.tensor(idx, dtype=torch.int64, device=dev)
                    iop_g = iop_all[it].contiguous()
                    for c0 in range(0, len(idx), DECODER_CHUNK):      # bounded decoder workspace however many components
                        c1 = min(c0 + DECODER_CHUNK, len(idx))
                        m_g, i_g, _ = sam.mask_decoder.predict_masks_tokens(
                            feat_tok, pe["pe_tok"], tokens[c0:c1].contiguous(), pe["no_mask"],
                            img_of_prompt=iop_g[c0:c1].contiguous())
                        masks[it[c0:c1]] = m_g
                        iou[it[c0:c1]] = i_g
            sel = 0 if self.use_cca else 1                                      # multimask_output = not use_cca; index 0
            _mark("prompt encoder + mask decoder")
            iou_host = iou[:, sel].cpu().numpy()
            for (b, start, cnt) in spans:
                # 7. upsample -> > 0 -> union over components -> nearest to the input size   ProtoSAM.py:669-676
                out = ops.mask_union(masks[start:start + cnt], sel, S, original_size, sam.variant_id(),
                                     sam.mask_threshold)
                results[b] = (out, [np.float32(v) for v in iou_host[start:start + cnt]])
            _mark("post-processing (upsample, threshold, union)")
            if B == 1:
                self.last_stats.update(low_res=masks, iou=iou, sel=sel)
            else:
                self.last_stats.update(low_res=masks, iou=iou, sel=sel, spans=spans)
        return results

    def _coarse_only_batch(self, output_logits, original_size):
        """`_coarse_only` for B slices at once: one softmax / argmax launch, one connected-components chain, one copy of the
        component tables to the host (round 5: forward_batch used to run SAM and drop its result when `coarse_pred_only` was set)."""
        B = output_logits.shape[0]
        if B == 1:
            return [self._coarse_only(output_logits, original_size)]
        dev = output_logits.device
        H = int(original_size)
        key = (str(dev), H, B)
        cache = self.__dict__.setdefault("_coarse_bufs", {})
        if key not in cache:
            while len(cache) >= 6:                   # (one set of buffers per batch size seen: keep the most recent few)
                cache.pop(next(iter(cache)))
            cache[key] = dict(fg_sum=torch.zeros(B, dtype=torch.int32, device=dev),
                              prob=torch.empty((B, 2, H, H), dtype=torch.float32, device=dev),
                              pred=torch.empty((B, H, H), dtype=torch.uint8, device=dev),
                              ccl=ops.CclWorkspace(H, H, MAX_COMPONENTS, dev, slots=B))
        bufs = cache[key]
        bufs["fg_sum"].zero_()
        prob, pred = ops.prob_argmax(output_logits.float().contiguous(), H, H, prob=bufs["prob"], pred=bufs["pred"],
                                     fg_sum=bufs["fg_sum"])
        cw = ops.ccl_batch(pred, prob, bufs["ccl"], fg_sum=bufs["fg_sum"])
        tabs = cw.tabs[:B].cpu().numpy()
        out = []
        for b in range(B):
            tab = tabs[b]
            if int(tab[0]) > int(tab[1]):                        # more components than the fast table holds: the one-slice path's large table
                out.append(self._coarse_only(output_logits[b:b + 1], original_size))
                continue
            n = int(tab[1])
            rows = tab[ops.CC_HDR:ops.CC_HDR + ops.CC_STRIDE * n].reshape(n, ops.CC_STRIDE)
            if not self.use_cca:
                out.append((pred[b].long(), [float(rows[:, 7].sum())]))
            elif n == 0:
                out.append((pred[b].long() * 0, [0]))
            else:
                k = int(tab[3])
                out.append(((cw.labels_b[b].view(H, H) == (k + 1)).long(), [float(rows[k, 7])]))
        return out

    def _coarse_only(self, output_logits, original_size):
        """ProtoSAM.py:580-590 (inference): logits (bilinear to the query's size if they differ) -> argmax map, mean fg
        confidence (get_confidence_from_logits); optional CCA keeps the best component and reports its confidence."""
        dev = output_logits.device
        H = int(original_size)
        key = (str(dev), H)
        cache = self.__dict__.setdefault("_coarse_bufs", {})     # (also called with a ProtoMedSAM as `self`)
        if key not in cache:
            cache[key] = dict(fg_sum=torch.zeros(1, dtype=torch.int32, device=dev),
                                   prob=torch.empty((1, 2, H, H), dtype=torch.float32, device=dev),
                                   pred=torch.empty((1, H, H), dtype=torch.uint8, device=dev),
                                   ccl=ops.CclWorkspace(H, H, MAX_COMPONENTS, dev, slots=1))
        bufs = cache[key]
        bufs["fg_sum"].zero_()
        prob, pred = ops.prob_argmax(output_logits.float().contiguous(), H, H, prob=bufs["prob"], pred=bufs["pred"],
                                     fg_sum=bufs["fg_sum"])
        cw = ops.ccl(pred[0], prob[0, 1], bufs["ccl"], fg_sum=bufs["fg_sum"])
        tab = cw.tab.cpu().numpy()
        if int(tab[0]) > int(tab[1]):                            # more components than the fast table holds: see forward_batch
            bufs["ccl_big"] = bufs.get("ccl_big") or ops.CclWorkspace(H, H, MAX_COMPONENTS_LARGE, dev, slots=1)
            cw = ops.ccl(pred[0], prob[0, 1], bufs["ccl_big"], fg_sum=bufs["fg_sum"])
            tab = cw.tab.cpu().numpy()
            if int(tab[0]) > int(tab[1]):
                raise RuntimeError(f"{int(tab[0])} connected components exceed the table capacity {MAX_COMPONENTS_LARGE}")
        n = int(tab[1])
        rows = tab[ops.CC_HDR:ops.CC_HDR + ops.CC_STRIDE * n].reshape(n, ops.CC_STRIDE)
        if not self.use_cca:
            return pred[0].long(), [float(rows[:, 7].sum())]
        if n == 0:
            return pred[0].long() * 0, [0]
        k = int(tab[3])
        keep = (cw.labels.view(H, H) == (k + 1)).long()
        return keep, [float(rows[k, 7])]
