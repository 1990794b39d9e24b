"""BASELINE config 2 (DINOv2 ViT-B/14 + ALP prototype match, coarse prediction only, 512x512) alone, for rocprofv3:
   python tools/config2_profile.py [batched|per_slice]   (bench.other_configs' two config-2 legs; 8 slices, 6 repetitions)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protosam_amd.runner import build_protosam, part_assign, run_slices, support_set
from protosam_amd.synth import synth_volume
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "batched"
m2, _ = build_protosam(dev, sam_type="vit_b", image_size=512, seed=1234, sam_depth=1, coarse_pred_only=True)
vol, _ = synth_volume(32, 512, seed=0, kind="mri")
svol, slab = synth_volume(32, 512, seed=1, kind="mri")
vol_d = vol.to(dev)
sup_imgs, sup_masks = support_set(svol.to(dev), slab.to(dev))
zs = [z for z in range(32) if part_assign(z, 32) == 1][:8]
batch = 8 if which == "batched" else 1
fn = lambda: run_slices(m2, vol_d, sup_imgs, sup_masks, zs, dev, batch=batch)   # noqa: E731
fn(); fn()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(6):
    fn()
torch.cuda.synchronize()
dt = time.perf_counter() - t
print(f"config2 {which}: {6 * 8 / dt:.1f} slices/s, {dt / 6 * 1e3:.2f} ms per 8 slices")
