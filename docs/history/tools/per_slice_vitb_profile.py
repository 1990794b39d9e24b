"""One ProtoSAM.forward per slice with SAM ViT-B (config 3's per-slice leg) or coarse only (config 2): wall per slice; under rocprofv3 the
GPU-busy share (tools/busy_share.py). python3 tools/per_slice_vitb_profile.py [full|coarse]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protosam_amd.runner import build_protosam, run_slices, support_set, part_assign
from protosam_amd.synth import synth_volume
mode = sys.argv[1] if len(sys.argv) > 1 else "full"
dev = torch.device("cuda:0")
kw = dict(sam_depth=1, coarse_pred_only=True) if mode == "coarse" else {}
m, _ = build_protosam(dev, sam_type="vit_b", image_size=512, seed=1234, **kw)
vol, _ = synth_volume(32, 512, seed=0, kind="mri"); svol, slab = synth_volume(32, 512, seed=1, kind="mri")
vol_d = vol.to(dev); sup_imgs, sup_masks = support_set(svol.to(dev), slab.to(dev))
zs = [z for z in range(32) if part_assign(z, 32) == 1][:8]
for _ in range(3):
    run_slices(m, vol_d, sup_imgs, sup_masks, zs, dev, batch=1)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(6):
    run_slices(m, vol_d, sup_imgs, sup_masks, zs, dev, batch=1)
torch.cuda.synchronize()
dt = time.perf_counter() - t
print(f"{mode}: {48 / dt:.1f} slices/s, {dt / 48 * 1e3:.3f} ms per slice")
