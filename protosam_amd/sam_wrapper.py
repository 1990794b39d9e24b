"""`SamWrapper` (models/SamWrapper.py:8-54): SAM's automatic mask generator as an oracle-guided coarse model - of all the
masks SAM proposes for an image, return the one with the best IoU against a given binary label.

Same constructor (`sam_args = {"model_type", "sam_checkpoint"}`) and `forward(image uint8 HWC, image_labels) -> bool
[H, W]` as the reference. The registry of the vendored package builds `SamBatched` (build_sam.py:66), whose
`postprocess_masks` interpolates with align_corners=True (modeling/sam.py:313-320), so that variant is selected here.

The reference downloads every proposed mask and scores it in numpy (:41-48). Here the masks are binarised on the device
together with their {tp, fp, fn} counts against the label; one mask crosses PCIe.

`sam_checkpoint = "random:<seed>[:<encoder_depth>]"` builds seeded synthetic weights (no checkpoints exist offline).
"""
import numpy as np
import torch
import torch.nn as nn

from .segment_anything import SamAutomaticMaskGenerator, sam_model_registry
from .segment_anything.utils.transforms import ResizeLongestSide


def get_iou(mask, label):
    """models/SamWrapper.py:8-13 (host version, for callers that hold numpy masks)."""
    tp = (mask * label).sum()
    fp = (mask * (1 - label)).sum()
    fn = ((1 - mask) * label).sum()
    return tp / (tp + fp + fn)


class SamWrapper(nn.Module):
    def __init__(self, sam_args):
        super().__init__()
        ckpt = sam_args["sam_checkpoint"]
        if isinstance(ckpt, str) and ckpt.startswith("random:"):
            from .synth import synth_state_dict
            parts = ckpt.split(":")
            depth = int(parts[2]) if len(parts) > 2 else None
            self.sam = sam_model_registry[sam_args["model_type"]](encoder_depth=depth)
            self.sam.load_state_dict(synth_state_dict(self.sam, int(parts[1])))
        else:
            self.sam = sam_model_registry[sam_args["model_type"]](checkpoint=ckpt)
        self.sam.postprocess_variant = "batched"
        self.sam.requires_grad_(False)
        self.mask_generator = SamAutomaticMaskGenerator(self.sam, **sam_args.get("generator_args", {}))
        self.transform = ResizeLongestSide(self.sam.image_encoder.img_size)
        self.last_stats = {}

    @torch.no_grad()
    def forward(self, image, image_labels, return_device=False):
        """image: HWC uint8; image_labels: binary [H, W] (numpy or tensor). -> bool numpy [H, W]: the proposal with the
        largest IoU against the label (the first one on ties, :44-46). `return_device=True` keeps it on the device as
        uint8 (used by `SamWrapperWrapper`)."""
        image = self.transform.apply_image(image)                               # :37
        dev = self.sam.device
        lab = torch.as_tensor(np.asarray(image_labels) if not torch.is_tensor(image_labels) else image_labels)
        if lab.numel() and (int(lab.min()) < 0 or int(lab.max()) > 1):
            raise ValueError("image_labels must be binary {0, 1} (models/ProtoSAM.py:124-125)")
        if tuple(lab.shape) != tuple(image.shape[:2]):
            raise ValueError(f"operands could not be broadcast together with shapes {tuple(image.shape[:2])} "
                             f"{tuple(lab.shape)}")                            # numpy's error in get_iou, :8-10
        lab = lab.to(device=dev, dtype=torch.uint8).contiguous()
        cand, masks, counts = self.mask_generator.generate_device(image, lab)   # :38
        if masks.shape[0] == 0:
            raise TypeError("list indices must be integers or slices, not NoneType")   # masks[None], :48
        c = counts.cpu().numpy().astype(np.float64)
        with np.errstate(divide="ignore", invalid="ignore"):
            iou = c[:, 0] / (c[:, 0] + c[:, 1] + c[:, 2])
        best, best_iou = None, 0                                                # :40-46
        for i, v in enumerate(iou):
            if best is None or v > best_iou:
                best, best_iou = i, v
        self.last_stats = dict(n_masks=int(masks.shape[0]), best_index=best, best_iou=float(best_iou), ious=iou,
                               candidates=cand)
        if return_device:
            return masks[best]
        return masks[best].cpu().numpy().astype(bool)

    def to(self, device):
        self.sam.to(device)
        return self

    def cuda(self, device=None):
        return self.to("cuda" if device is None else device)
