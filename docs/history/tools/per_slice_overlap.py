import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from protosam_amd.runner import build_protosam, run_slices, support_set, part_assign
from protosam_amd.synth import synth_volume
dev = torch.device("cuda:0")
model, _ = build_protosam(dev, sam_type="vit_h", image_size=512, seed=1234)
vol, lab = synth_volume(64, 512, seed=0, kind="ct")
svol, slab = synth_volume(64, 512, seed=1, kind="ct")
vol_d = vol.to(dev)
sup_imgs, sup_masks = support_set(svol.to(dev), slab.to(dev))
zs = [z for z in range(64) if part_assign(z, 64) == 1][:20]
res = {}
for rep in range(2):
    for mode in ("0", "1"):
        model.overlap_streams = mode
        run_slices(model, vol_d, sup_imgs, sup_masks, zs[:2], dev, batch=1)
        torch.cuda.synchronize(); t = time.perf_counter()
        out, st = run_slices(model, vol_d, sup_imgs, sup_masks, zs, dev, batch=1)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        res[mode] = out.clone()
        print(f"overlap={mode}: {len(zs)/dt:.2f} slices/s", flush=True)
print("identical:", torch.equal(res["0"], res["1"]))
