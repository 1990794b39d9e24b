"""Per-slice overlap metrics of the caller (validation_protosam.py:169-185 `get_dice_iou_precision_recall`).

Evaluation bookkeeping on two [H, W] {0,1} maps, not a stage of the hot path: the reference computes it on the host
after `query_pred.cpu()` (validation_protosam.py:389,400). Same formulas, same 1e-8 guards, same all-background early
return (a dict WITHOUT the "iou" key, as the reference). Runs on whatever device the two tensors live on.
bench.py, __graft_entry__.smoke() and the parity tests use `dice` as the "Dice-equivalent" agreement measure.
"""
import torch


def get_dice_iou_precision_recall(pred: torch.Tensor, gt: torch.Tensor):
    if gt.sum() == 0:
        print("gt is all background")
        return {"dice": 0, "precision": 0, "recall": 0}
    tp = (pred * gt).sum()
    fp = (pred * (1 - gt)).sum()
    fn = ((1 - pred) * gt).sum()
    dice = 2 * tp / (2 * tp + fp + fn + 1e-8)
    precision = tp / (tp + fp + 1e-8)
    recall = tp / (tp + fn + 1e-8)
    iou = tp / (tp + fp + fn + 1e-8)
    return {"dice": dice, "iou": iou, "precision": precision, "recall": recall}


def dice(pred, gt):
    """float Dice of two {0,1} maps (1.0 when both are empty: identical masks agree)."""
    pred, gt = torch.as_tensor(pred).float(), torch.as_tensor(gt).float()
    if gt.sum() == 0:
        return 1.0 if pred.sum() == 0 else 0.0
    return float(get_dice_iou_precision_recall(pred, gt)["dice"])
