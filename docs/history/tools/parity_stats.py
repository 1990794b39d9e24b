"""Distribution of |sigmoid(low_res) - oracle| over every slice of config 3 / 4 on the batched (16-slice, folded LayerNorm) path
and on the one-slice-per-call path (PSAM_STATS_BATCH=1): max, 99.99th / 99.9th percentile, mean - to tell a shift of the error level
from a reshuffle of its worst pixel; per-slice Dice / flipped pixels of the final masks beside it.
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
    BATCH = int(os.environ.get("PSAM_STATS_BATCH", "16"))
    from protosam_amd.metrics import dice
    dices, flips = [], []
    for i in range(0, n, BATCH):
        chunk = list(range(n))[i:i + BATCH]
        masks, _ = run_slices(model, vol_d, sup_imgs, sup_masks, chunk, dev, batch=BATCH)
        per = model.last_stats
        if "low_res" not in per:
            continue
        low, sel = per["low_res"].cpu(), per["sel"]
        if BATCH == 1:
            per = dict(per, spans=[(0, 0, low.shape[0])] if low.shape[0] else [])
        for b, z in enumerate(chunk):
            refm = T._unpack(gold[f"z{z}_mask"], 512)
            dices.append(dice(masks[b].cpu().float(), refm))
            flips.append(int((masks[b].cpu().float() != refm).sum()))
            sp = [s for s in per["spans"] if s[0] == b]
            if not sp:
                continue
            _, start, cnt = sp[0]
            refp = torch.from_numpy(gold[f"z{z}_prob4"].astype(np.float32) / 65535.0)
            p = torch.sigmoid(low[start:start + cnt, sel])[..., ::4, ::4]
            errs.append((p - refp).abs().flatten())
    e = torch.cat(errs).double()
    q = torch.quantile(e[torch.randperm(e.numel())[:4000000]], torch.tensor([0.999, 0.9999], dtype=torch.float64))
    print(f"  final masks: mean Dice {np.mean(dices):.5f}, worst {min(dices):.5f} (slice {int(np.argmin(dices))}), most flipped {max(flips)}, slices below 0.999: {sum(d < 0.999 for d in dices)}, "
          f"below 0.998 with more than 32 flips: {[(z, round(d, 5), f) for z, (d, f) in enumerate(zip(dices, flips)) if d < 0.998 and f > 32]}")
    print(f"config {cfg} batch {BATCH}: {e.numel()} pixels, max {e.max().item():.3e}, p99.99 {q[1].item():.3e}, p99.9 {q[0].item():.3e}, mean {e.mean().item():.3e}, "
          f"pixels above 5e-4: {(e > 5e-4).sum().item()}")
