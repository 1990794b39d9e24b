"""BASELINE config 3 (DINOv2 ViT-B/14 + ALP + SAM ViT-B, 32-slice MRI-like volume, two 16-slice batches) alone, for rocprofv3:
  rocprofv3 --kernel-trace --stats -d out -o c3 -- python3 tools/config3_profile.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protosam_amd.runner import build_protosam, run_slices, support_set
from protosam_amd.synth import synth_volume
dev = torch.device("cuda:0")
m3, _ = build_protosam(dev, sam_type="vit_b", image_size=512, seed=1234)
vol, _ = synth_volume(32, 512, seed=0, kind="mri")
svol, slab = synth_volume(32, 512, seed=1, kind="mri")
vol_d = vol.to(dev)
sup_imgs, sup_masks = support_set(svol.to(dev), slab.to(dev))
zs = list(range(32))
m3.overlap_streams = "0"
for _ in range(2):
    run_slices(m3, vol_d, sup_imgs, sup_masks, zs, dev, batch=16)
torch.cuda.synchronize()
t = time.perf_counter()
reps = 4
for _ in range(reps):
    run_slices(m3, vol_d, sup_imgs, sup_masks, zs, dev, batch=16)
torch.cuda.synchronize()
dt = time.perf_counter() - t
print(f"config 3: {reps * 32 / dt:.1f} slices/s, {dt / reps / 2 * 1e3:.2f} ms per 16-slice batch")
