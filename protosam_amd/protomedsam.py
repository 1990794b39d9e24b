"""`ProtoMedSAM` on the HIP path (mirror of /root/reference/models/ProtoMedSAM.py:10-222).

Same constructor and `forward(query_image, coarse_model_input, degrees_rotate=0) -> (uint8 mask [H,W], [conf])` contract.
Differences from `ProtoSAM` that are reproduced: box prompts only; the image is min-max scaled to [0,1]
floats (no uint8 quantisation, no mean/std) before `medsam.image_encoder`; `mask_decoder(multimask_output=False)`;
`sigmoid` BEFORE the bilinear resize and a 0.5 threshold; the component confidences are computed from
`softmax(softmax(logits))` because `get_connected_components` re-applies softmax to the already-softmaxed tensor
(ProtoMedSAM.py:178-187, util/utils.py:485); an empty coarse mask returns the arg-max map at the ORIGINAL size (:194-197).
More than one connected component without `use_cca` is undefined in the reference (5-D nearest interpolate, SURVEY
Q16) and raises here.
"""
import numpy as np
import torch
import torch.nn as nn

from . import ops
from .protosam import MAX_COMPONENTS, MAX_COMPONENTS_LARGE, ModelWrapper
from .segment_anything import sam_model_registry


class ProtoMedSAM(nn.Module):
    def __init__(self, image_size, coarse_segmentation_model: ModelWrapper,
                 sam_pretrained_path="pretrained_model/medsam_vit_b.pth", debug=False, use_cca=False,
                 coarse_pred_only=False):
        super().__init__()
        if isinstance(image_size, int):
            image_size = (image_size, image_size)
        self.image_size = image_size
        self.coarse_segmentation_model = coarse_segmentation_model
        self.get_sam(sam_pretrained_path)
        self.coarse_pred_only = coarse_pred_only
        self.debug = debug
        self.use_cca = use_cca
        if debug:
            raise NotImplementedError("debug plots are outside the accelerated path")
        if tuple(self.image_size) != (1024, 1024):
            raise NotImplementedError("image_size must be (1024, 1024) as in validation_protosam.py:234")
        self._ccl = None
        self._bufs = {}
        self.last_stats = {}

    def get_sam(self, checkpoint_path):
        model_type = "vit_b"
        if checkpoint_path is not None and "vit_h" in checkpoint_path:
            model_type = "vit_h"
        if checkpoint_path is not None and checkpoint_path.startswith("random:"):
            from .synth import synth_state_dict
            parts = checkpoint_path.split(":")
            model_type = parts[1]
            seed = int(parts[2]) if len(parts) > 2 else 1234
            depth = int(parts[3]) if len(parts) > 3 else None
            self.medsam = sam_model_registry[model_type](encoder_depth=depth)
            self.medsam.load_state_dict(synth_state_dict(self.medsam, seed))
            self.medsam.eval()
        else:
            self.medsam = sam_model_registry[model_type](checkpoint=checkpoint_path).eval()

    # ---- the reference's helper methods by name (ProtoMedSAM.py:31-120,224-249); `forward` does their work inside fused kernels ----
    @torch.no_grad()
    def medsam_inference(self, img_embed, box_1024, H, W, query_label=None):
        """ProtoMedSAM.py:31-66: box prompts [B,4] (XYXY in the 1024 frame) on ONE image embedding [1,256,64,64] -> (uint8 masks as
        numpy - `squeeze()`d like the reference: [H,W] for one box - , conf numpy [B,C]); sigmoid BEFORE the bilinear resize, 0.5
        threshold. With `query_label` the decoder returns its three masks and the one with the best IoU against the label is kept
        (`get_best_mask`), as [1,H,W]."""
        sam = self.medsam
        box_torch = torch.as_tensor(box_1024, dtype=torch.float, device=img_embed.device)
        if len(box_torch.shape) == 2:
            box_torch = box_torch[:, None, :]                                           # (B, 1, 4)
        sparse, dense = sam.prompt_encoder(points=None, boxes=box_torch, masks=None)
        low, conf = sam.mask_decoder(image_embeddings=img_embed, image_pe=sam.prompt_encoder.get_dense_pe(),
                                     sparse_prompt_embeddings=sparse, dense_prompt_embeddings=dense,
                                     multimask_output=True if query_label is not None else False)
        if H != W:
            raise NotImplementedError("medsam_inference: square outputs (psam_mask_union)")
        low = low.contiguous()
        n, C = low.shape[:2]
        segs = torch.stack([ops.mask_union(low[i:i + 1], c, int(H), int(H), 3, 0.5) for i in range(n) for c in range(C)])
        medsam_seg = segs.view(n, C, H, W).squeeze().to(torch.uint8).cpu().numpy()
        if query_label is not None:
            medsam_seg = self.get_best_mask(medsam_seg, query_label)[None, :]
        return medsam_seg, conf.cpu().detach().numpy()

    def get_iou(self, pred, label):
        """ProtoMedSAM.py:68-77 (numpy uint8 [h,w] each)."""
        tp = np.logical_and(pred, label).sum()
        fp = np.logical_and(pred, 1 - label).sum()
        fn = np.logical_and(1 - pred, label).sum()
        return tp / (tp + fp + fn)

    def get_best_mask(self, masks, labels):
        """ProtoMedSAM.py:79-92: masks numpy [B,h,w], labels tensor [1,H,W] -> the mask with the largest IoU (None when every IoU is 0)."""
        np_labels = labels[0].clone().detach().cpu().numpy()
        best_iou, best_mask = 0, None
        for mask in masks:
            iou = self.get_iou(mask, np_labels)
            if iou > best_iou:
                best_iou, best_mask = iou, mask
        return best_mask

    def get_bbox(self, pred):
        """ProtoMedSAM.py:94-106: [xmin, ymin, xmax, ymax] of the non-zero pixels of pred [H,W] (tensor or numpy), None when empty."""
        if isinstance(pred, np.ndarray):
            pred = torch.from_numpy(pred)
        if pred.max() == 0:
            return None
        indices = torch.nonzero(pred)
        ymin, xmin = indices.min(dim=0)[0]
        ymax, xmax = indices.max(dim=0)[0]
        return np.array([int(xmin), int(ymin), int(xmax), int(ymax)])

    def get_bbox_per_cc(self, conn_components):
        """ProtoMedSAM.py:109-120: one XYXY box per label 1 ... n-1 of a cv2-style `(n, labels, stats, centroids)` tuple
        (protosam_amd.utils.cca / get_connected_components)."""
        return np.array([self.get_bbox(torch.as_tensor(np.asarray(conn_components[1]) == i).to(torch.uint8))
                         for i in range(1, conn_components[0])])

    @torch.no_grad()
    def segment_all(self, query_image, query_label):
        """ProtoMedSAM.py:224-249: the whole-image box [0,0,W,H] as the prompt, three masks, the one closest to `query_label` kept.
        query_image [1,3,1024,1024] on the device (the reference hands it to the image encoder as it is)."""
        H, W = query_image.shape[-2:]
        sam = self.medsam
        S = sam.image_encoder.img_size
        if (H, W) != (S, S):
            raise ValueError(f"segment_all: the image encoder takes {S} x {S} images, got {(H, W)}")      # (the reference's encoder asserts)
        bbox = np.array([[0, 0, W, H]])
        bufs = self._work(query_image.device)
        q = query_image.float().contiguous()
        ops.minmax(q, 1, mm=bufs["mm"])                                                  # (x - min) / (max - min), :230
        ops.sam_patchify(q, bufs["mm"], S, sam.image_encoder.patch_size, (0.0, 0.0, 0.0), (1.0, 1.0, 1.0), False, out=bufs["patches"])
        g = sam.image_encoder.grid
        emb = sam.image_encoder.encode_patches(bufs["patches"], 1).view(1, g, g, -1).permute(0, 3, 1, 2)
        medsam_seg, conf = self.medsam_inference(emb, bbox, H, W, query_label)
        medsam_seg = torch.tensor(medsam_seg, device=query_image.device)
        return medsam_seg.view(H, W), [conf]

    def _work(self, dev):
        if not self._bufs:
            self._bufs = dict(fg_sum=torch.zeros(1, dtype=torch.int32, device=dev),
                              prob=torch.empty((1, 2, 1024, 1024), dtype=torch.float32, device=dev),
                              prob2=torch.empty((1, 2, 1024, 1024), dtype=torch.float32, device=dev),
                              pred=torch.empty((1, 1024, 1024), dtype=torch.uint8, device=dev),
                              pred2=torch.empty((1, 1024, 1024), dtype=torch.uint8, device=dev),
                              q1024=torch.empty((1, 3, 1024, 1024), dtype=torch.float32, device=dev),
                              mm=torch.empty(2, dtype=torch.int32, device=dev),
                              patches=torch.empty((4096, 768), dtype=torch.float16, device=dev),
                              event=torch.cuda.Event())
            self._ccl = ops.CclWorkspace(1024, 1024, MAX_COMPONENTS, dev)
        return self._bufs

    @torch.no_grad()
    def forward(self, query_image, coarse_model_input, degrees_rotate=0):
        from .protosam import ProtoSAM
        original_size = query_image.shape[-2]
        dev = query_image.device
        # ProtoMedSAM.py:129-139 (rotation TTA around the coarse model; identity at 0 degrees)
        output_logits = ProtoSAM._coarse_logits(self, query_image, coarse_model_input, degrees_rotate)   # [1,2,H,W] ALP logits
        if self.coarse_pred_only:                                                          # ProtoMedSAM.py:163-172
            return ProtoSAM._coarse_only(self, output_logits, original_size)
        bufs = self._work(dev)
        sam = self.medsam
        S = sam.image_encoder.img_size
        bufs["fg_sum"].zero_()
        # bilinear to 1024 -> softmax (need_softmax is True for ALP logits, :178-179) -> argmax
        output_p, pred = ops.prob_argmax(output_logits.float().contiguous(), S, S, prob=bufs["prob"], pred=bufs["pred"],
                                         fg_sum=bufs["fg_sum"])
        # get_connected_components softmaxes its `logits` argument again (util/utils.py:485)
        p2, _ = ops.prob_argmax(output_p, S, S, prob=bufs["prob2"], pred=bufs["pred2"])
        cw = ops.ccl(pred[0], p2[0, 1], self._ccl, fg_sum=bufs["fg_sum"])
        cw.tab_host.copy_(cw.tab, non_blocking=True)
        bufs["event"].record()
        q = query_image.float().contiguous()
        if tuple(q.shape[-2:]) != (S, S):
            q = ops.bilinear_nchw(q, S, S, out=bufs["q1024"])
        ops.minmax(q, 1, mm=bufs["mm"])
        ops.sam_patchify(q, bufs["mm"], S, sam.image_encoder.patch_size, (0.0, 0.0, 0.0), (1.0, 1.0, 1.0), False,
                         out=bufs["patches"])                                              # :203-205
        feat_tok = sam.image_encoder.encode_patches(bufs["patches"], 1)[0]
        bufs["event"].synchronize()
        tab = cw.tab_host.numpy()
        if int(tab[0]) > int(tab[1]):   # more components than the fast table holds (cca searches ALL of them, utils.py:496-541)
            if getattr(self, "_ccl_big", None) is None:
                self._ccl_big = ops.CclWorkspace(1024, 1024, MAX_COMPONENTS_LARGE, dev)
            tab = ops.ccl(pred[0], p2[0, 1], self._ccl_big, fg_sum=bufs["fg_sum"]).tab.cpu().numpy()
            if int(tab[0]) > int(tab[1]):
                raise RuntimeError(f"{int(tab[0])} connected components exceed the table capacity {MAX_COMPONENTS_LARGE}")
        n = int(tab[1])
        self.last_stats = dict(n_components=int(tab[0]))
        if n == 0:                                                                         # :194-197
            return torch.zeros((original_size, original_size), dtype=torch.int64, device=dev), [0]
        rows = tab[ops.CC_HDR:ops.CC_HDR + ops.CC_STRIDE * n].reshape(n, ops.CC_STRIDE)
        if self.use_cca:
            rows = rows[int(tab[3]):int(tab[3]) + 1]
        if rows.shape[0] != 1:
            raise NotImplementedError("ProtoMedSAM with several components is undefined in the reference (SURVEY Q16); "
                                      "use use_cca=True")
        boxes = rows[:, 3:7] / np.array([S, S, S, S]) * max(self.image_size)               # :201-202 (identity at 1024)
        coords = boxes.reshape(-1, 2, 2).astype(np.float32)
        labels = np.tile(np.array([[2, 3]], dtype=np.int32), (coords.shape[0], 1))
        pe = sam.prompt_encoder._packed()
        dpk = sam.mask_decoder._packed()
        tokens = ops.prompt_tokens(torch.from_numpy(coords).to(dev), torch.from_numpy(labels).to(dev), pe["G"],
                                   pe["type_emb"], dpk["out_tok"], coords.shape[0], 2, float(S))
        masks, iou, _ = sam.mask_decoder.predict_masks_tokens(feat_tok, pe["pe_tok"], tokens, pe["no_mask"])
        seg = ops.mask_union(masks, 0, S, original_size, 3, 0.5)                           # sigmoid -> bilinear -> > 0.5
        self.last_stats.update(low_res=masks, iou=iou)
        return seg.to(torch.uint8), [iou[:, 0:1].cpu().numpy()]

    @torch.no_grad()
    def forward_classes(self, query_image, support_image, support_masks, val_wsize=2):
        """Throughput path of the multi-class loop (BASELINE config 5; /root/reference/validation.py:207 runs one 1-way episode per
        class on the same slice): ONE DINOv2 forward of the query (and of the support) shared by all classes' prototype banks, ONE
        MedSAM image-encoder forward of the query, one batched box-prompt decoder call for all classes. Returns a list of
        (uint8 mask [H,W], [conf]) per class, equal to `forward()` of each class alone (same arithmetic; the decoder batches).
        Only `use_cca=True` (one component per class), as `forward()`."""
        if not self.use_cca or self.coarse_pred_only:
            raise NotImplementedError("forward_classes: use_cca=True, coarse_pred_only=False")
        original_size = query_image.shape[-2]
        dev = query_image.device
        alp = self.coarse_segmentation_model.model
        logits = alp.forward_classes(support_image, support_masks, query_image, isval=True, val_wsize=val_wsize)
        bufs = self._work(dev)
        sam = self.medsam
        S = sam.image_encoder.img_size
        nc = len(logits)
        if len(getattr(self, "_cls_ws", [])) < nc:
            self._cls_ws = [dict(ccl=ops.CclWorkspace(1024, 1024, MAX_COMPONENTS, dev), fg=torch.zeros(1, dtype=torch.int32, device=dev),
                                 prob=torch.empty((1, 2, 1024, 1024), dtype=torch.float32, device=dev),
                                 pred=torch.empty((1, 1024, 1024), dtype=torch.uint8, device=dev)) for _ in range(nc)]
        cws = []
        for c, lg in enumerate(logits):
            w = self._cls_ws[c]
            w["fg"].zero_()
            output_p, pred = ops.prob_argmax(lg.float().contiguous(), S, S, prob=w["prob"], pred=w["pred"], fg_sum=w["fg"])
            p2, _ = ops.prob_argmax(output_p, S, S, prob=bufs["prob2"], pred=bufs["pred2"])
            cw = ops.ccl(pred[0], p2[0, 1], w["ccl"], fg_sum=w["fg"])
            cw.tab_host.copy_(cw.tab, non_blocking=True)
            cws.append(cw)
        bufs["event"].record()
        q = query_image.float().contiguous()
        if tuple(q.shape[-2:]) != (S, S):
            q = ops.bilinear_nchw(q, S, S, out=bufs["q1024"])
        ops.minmax(q, 1, mm=bufs["mm"])
        ops.sam_patchify(q, bufs["mm"], S, sam.image_encoder.patch_size, (0.0, 0.0, 0.0), (1.0, 1.0, 1.0), False, out=bufs["patches"])
        feat_tok = sam.image_encoder.encode_patches(bufs["patches"], 1)[0]
        bufs["event"].synchronize()
        results = [None] * nc
        boxes, owners = [], []
        for c, cw in enumerate(cws):
            tab = cw.tab_host.numpy()
            if int(tab[0]) > int(tab[1]):
                raise RuntimeError(f"class {c}: {int(tab[0])} connected components exceed the fast table; use forward() for this slice")
            n = int(tab[1])
            if n == 0:
                results[c] = (torch.zeros((original_size, original_size), dtype=torch.int64, device=dev), [0])
                continue
            rows = tab[ops.CC_HDR:ops.CC_HDR + ops.CC_STRIDE * n].reshape(n, ops.CC_STRIDE)
            row = rows[int(tab[3])]
            boxes.append(row[3:7] / np.array([S, S, S, S]) * max(self.image_size))
            owners.append(c)
        self.last_stats = dict(n_classes=nc, n_prompted=len(owners))
        if owners:
            coords = np.stack(boxes).reshape(-1, 2, 2).astype(np.float32)
            labels = np.tile(np.array([[2, 3]], dtype=np.int32), (coords.shape[0], 1))
            pe = sam.prompt_encoder._packed()
            dpk = sam.mask_decoder._packed()
            tokens = ops.prompt_tokens(torch.from_numpy(coords).to(dev), torch.from_numpy(labels).to(dev), pe["G"], pe["type_emb"],
                                       dpk["out_tok"], coords.shape[0], 2, float(S))
            masks, iou, _ = sam.mask_decoder.predict_masks_tokens(feat_tok, pe["pe_tok"], tokens, pe["no_mask"])
            iou_h = iou[:, 0:1].cpu().numpy()
            for k, c in enumerate(owners):
                seg = ops.mask_union(masks[k:k + 1], 0, S, original_size, 3, 0.5)
                results[c] = (seg.to(torch.uint8).clone(), [iou_h[k:k + 1]])
            self.last_stats.update(low_res=masks, iou=iou)
        return results
