"""The functions of /root/reference/util/utils.py that sit on the hot path, by name, on the device kernels:
`get_connected_components` (:474-494), `cca` (:496-541), `get_confidence_from_logits` (:429-434), `need_softmax` (:60-61);
`rotate_tensor_no_crop` / `reverse_tensor` live in protosam_amd/rotate.py and are re-exported here.

`ProtoSAM.forward` never calls these (its component table stays on the device and only ~25 KB travel to the host); they exist
for callers that use the reference's helpers directly. The labelling is `psam_ccl` (csrc/ccl.hip); the cv2-style tuple
`(n_labels, labels int32 [H,W], stats int32 [n,5] = left, top, width, height, area, centroids float64 [n,2])` is assembled on the
host from the kernel's table. Labels are numbered by the raster order of each component's first pixel.
"""
import numpy as np
import torch

from . import ops
from .rotate import reverse_tensor, rotate_tensor_no_crop  # noqa: F401

_WS = {}
CAPACITY = 4096      # psam_ccl's table limit


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("protosam_amd.utils: connected components run on the GPU (psam_ccl); no CPU path")
    return torch.device("cuda", torch.cuda.current_device())


def get_confidence_from_logits(logits):
    """util/utils.py:429-434: mean foreground probability over the pixels predicted foreground."""
    p = logits.softmax(1)[:, 1].flatten(1)
    pred = (p >= 0.5).float()
    return ((p * pred).sum() / (pred.sum() + 1e-6)).item()


def need_softmax(tensor, dim=1):
    """util/utils.py:60-61 (the reference's expression, broadcasting included): True unless the values along `dim` already sum to one
    and none is negative."""
    return not torch.all(torch.isclose(tensor.sum(dim=dim), torch.ones_like(tensor.sum(dim=dim))) & (tensor >= 0))


def get_connected_components(query_pred_original, query_pred_logits, return_conf=False):
    """util/utils.py:474-494: 8-connected components of the binary map `query_pred_original` (numpy [H,W]) and, with `return_conf`,
    conf[j] = sum(softmax(logits)[:,1] * [label == j]) / (sum(pred) + 1e-6) per label (0 for the background).
    -> (cca_output, conf dict or None)."""
    pred_np = np.asarray(query_pred_original)
    H, W = pred_np.shape
    dev = query_pred_logits.device if (isinstance(query_pred_logits, torch.Tensor) and query_pred_logits.is_cuda) else _device()
    key = (str(dev), H, W)
    if key not in _WS:
        _WS.clear()                                          # one shape at a time (64 MB of scratch at 1024 x 1024)
        _WS[key] = ops.CclWorkspace(H, W, CAPACITY, dev, slots=1)
    ws = _WS[key]
    pred = torch.from_numpy(np.ascontiguousarray((pred_np != 0).astype(np.uint8))).to(dev)
    if return_conf or query_pred_logits is not None:
        probs = query_pred_logits.to(dev).float().softmax(1)[0, 1].contiguous()
    else:
        probs = torch.zeros((H, W), dtype=torch.float32, device=dev)
    fg_sum = torch.tensor([int(pred_np.astype(np.int64).sum())], dtype=torch.int32, device=dev)      # sum(pred), :490
    ops.ccl(pred, probs, ws, fg_sum=fg_sum)
    tab = ws.tab.cpu().numpy()
    if int(tab[0]) > int(tab[1]):
        raise RuntimeError(f"{int(tab[0])} connected components exceed the table capacity {CAPACITY} (csrc/ccl.hip)")
    n = int(tab[1])
    rows = tab[ops.CC_HDR:ops.CC_HDR + ops.CC_STRIDE * n].reshape(n, ops.CC_STRIDE)
    labels = ws.labels.view(H, W).cpu().numpy().astype(np.int32)
    stats = np.zeros((n + 1, 5), dtype=np.int32)
    cent = np.zeros((n + 1, 2), dtype=np.float64)
    bys, bxs = np.nonzero(labels == 0)                        # row 0: the background, as cv2 reports it
    if len(bys):
        stats[0] = (bxs.min(), bys.min(), bxs.max() - bxs.min() + 1, bys.max() - bys.min() + 1, len(bys))
        cent[0] = (bxs.sum(dtype=np.float64) / len(bys), bys.sum(dtype=np.float64) / len(bys))
    for k, r in enumerate(rows):
        stats[k + 1] = (r[3], r[4], r[5] - r[3] + 1, r[6] - r[4] + 1, r[0])
        cent[k + 1] = (r[1] / r[0], r[2] / r[0])
    cca_output = (n + 1, labels, stats, cent)
    if not return_conf:
        return cca_output, None
    conf = {0: 0}
    for k, r in enumerate(rows):
        conf[k + 1] = np.float32(r[7])
    return cca_output, conf


def cca(query_pred_original, query_pred_logits, return_conf=False, return_cc=False):
    """util/utils.py:496-541: keep the most confident connected component. `return_cc`: the cv2-style tuple reduced to background +
    that component (relabelled 1); `return_conf`: (pred restricted to it, its confidence); else the restricted pred."""
    cca_output, cca_conf = get_connected_components(query_pred_original, query_pred_logits, return_conf=True)
    max_conf, max_key = cca_conf[0], 0
    for k, v in cca_conf.items():
        if v > max_conf:
            max_conf, max_key = v, k
    pred_np = np.asarray(query_pred_original)
    if max_conf == 0:
        query_pred = np.zeros_like(pred_np)
    else:
        cca_output = (2, np.where(cca_output[1] != max_key, 0, 1), cca_output[2][[0, max_key]], cca_output[3][[0, max_key]])
        query_pred = (cca_output[1] == 1).astype(np.uint8)
    if return_cc:
        return cca_output
    out = pred_np * query_pred
    if return_conf:
        return out, max_conf
    return out
