"""Distribution of |sigmoid(low_res) - oracle| over every slice of config 3 / 4 on the batched (16-slice, folded LayerNorm) path:
max, 99.99th / 99.9th percentile, mean - to tell a shift of the error level from a reshuffle of its worst pixel.
  python tools/parity_stats.py [3|4]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import test_fullsize_gpu as T
from protosam_amd.runner import run_slices
dev = torch.device("cuda:0")
for cfg in ([int(a) for a in sys.argv[1:]] or [3, 4]):
    gold = np.load(os.path.join(T.GOLD, f"fullvolume_cfg{cfg}.npz"))
    model, vol_d, sup_imgs, sup_masks, n, _, _ = T._volume_setup(dev, cfg)
    model.use_cca = False
    errs = []
    for i in range(0, n, 16):
        chunk = list(range(n))[i:i + 16]
        run_slices(model, vol_d, sup_imgs, sup_masks, chunk, dev, batch=16)
        per = model.last_stats
        low, sel = per["low_res"].cpu(), per["sel"]
        for b, z in enumerate(chunk):
            sp = [s for s in per["spans"] if s[0] == b]
            if not sp:
                continue
            _, start, cnt = sp[0]
            refp = torch.from_numpy(gold[f"z{z}_prob4"].astype(np.float32) / 65535.0)
            p = torch.sigmoid(low[start:start + cnt, sel])[..., ::4, ::4]
            errs.append((p - refp).abs().flatten())
    e = torch.cat(errs).double()
    q = torch.quantile(e[torch.randperm(e.numel())[:4000000]], torch.tensor([0.999, 0.9999], dtype=torch.float64))
    print(f"config {cfg} batched: {e.numel()} pixels, max {e.max().item():.3e}, p99.99 {q[1].item():.3e}, p99.9 {q[0].item():.3e}, mean {e.mean().item():.3e}, "
          f"pixels above 5e-4: {(e > 5e-4).sum().item()}")
