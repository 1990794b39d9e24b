"""Times SamWrapper.forward / SamAutomaticMaskGenerator.generate (32x32 grid, SAM ViT-H unless told otherwise).

  python tools/bench_amg.py [--sam vit_h] [--depth N] [--iters 5] [--chunk 256] [--points 32]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sam", default="vit_h")
    ap.add_argument("--depth", type=int, default=None)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--chunk", type=int, default=256)
    ap.add_argument("--points", type=int, default=32)
    a = ap.parse_args()
    from protosam_amd.sam_wrapper import SamWrapper
    from protosam_amd.synth import synth_pair
    ck = "random:7" + (f":{a.depth}" if a.depth else "")
    # synthetic weights: thresholds 0 / no suppression keep every candidate alive, the worst case for the back end;
    # the defaults (0.88 / 0.95 / 0.7) keep almost none
    res = {}
    _, _, q, gt = synth_pair(1024, seed=3)
    q = q[0].permute(1, 2, 0).numpy()
    img = ((q - q.min()) / (q.max() - q.min()) * 255).astype(np.uint8)
    label = gt[0].numpy().astype(np.uint8)
    for name, gargs in (("defaults", {}),
                        ("keep_top64", dict(pred_iou_thresh=0.0, stability_score_thresh=0.0, box_nms_thresh=1.0))):
        w = SamWrapper({"model_type": a.sam, "sam_checkpoint": ck,
                        "generator_args": dict(points_per_side=a.points, decode_chunk=a.chunk, **gargs)}).cuda()
        g = w.mask_generator
        if name == "keep_top64":
            cand = g._candidates(img)
            thr = float(np.sort(cand["iou_preds"])[-64])
            g.pred_iou_thresh = thr if thr > 0 else 1e-9
        for phase in ("candidates", "wrapper"):
            fn = (lambda: g._candidates(img)) if phase == "candidates" else (lambda: w(img, label))
            try:
                fn()
            except TypeError:      # no proposal survived (defaults with synthetic weights)
                res[f"{name}_{phase}_ms"] = None
                continue
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(a.iters):
                fn()
            torch.cuda.synchronize()
            res[f"{name}_{phase}_ms"] = round((time.perf_counter() - t) / a.iters * 1e3, 2)
        res[f"{name}_n_masks"] = w.last_stats.get("n_masks")
        del w
        torch.cuda.empty_cache()
    res.update(sam=a.sam, depth=a.depth, points=a.points ** 2, chunk=a.chunk)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
