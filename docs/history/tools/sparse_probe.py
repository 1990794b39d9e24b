import sys, torch
sys.path.insert(0, "/root/repo")
from protosam_amd.runner import build_protosam, run_slices, support_set
from protosam_amd.synth import synth_volume
dev = torch.device("cuda:0")
model, _ = build_protosam(dev, sam_type="vit_b", image_size=512, seed=1234, sam_depth=2)
svol, slab = synth_volume(64, 512, seed=1, kind="ct")
sup_imgs, sup_masks = support_set(svol.to(dev), slab.to(dev))
vol = synth_volume(64, 512, seed=0, kind="ct_sparse")[0].to(dev)
masks, st = run_slices(model, vol, sup_imgs, sup_masks, list(range(64)), dev, batch=16)
print("prompt sets per slice:", st)
print("mask px:", [int(m.sum()) for m in masks][::4])
