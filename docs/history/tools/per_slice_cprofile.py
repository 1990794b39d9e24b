"""cProfile of the host side of one ProtoSAM.forward per slice (SAM ViT-B: the launch-/host-bound case). Top cumulative entries."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protosam_amd.runner import build_protosam, run_slices, support_set, part_assign
from protosam_amd.synth import synth_volume
dev = torch.device("cuda:0")
m, _ = build_protosam(dev, sam_type="vit_b", image_size=512, seed=1234)
vol, _ = synth_volume(32, 512, seed=0, kind="mri"); svol, slab = synth_volume(32, 512, seed=1, kind="mri")
vol_d = vol.to(dev); sup_imgs, sup_masks = support_set(svol.to(dev), slab.to(dev))
zs = [z for z in range(32) if part_assign(z, 32) == 1][:8]
for _ in range(3):
    run_slices(m, vol_d, sup_imgs, sup_masks, zs, dev, batch=1)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(6):
    run_slices(m, vol_d, sup_imgs, sup_masks, zs, dev, batch=1)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
