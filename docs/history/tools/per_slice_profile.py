"""One ProtoSAM.forward per slice (the reference-shaped call, validation_protosam.py:387): wall time per slice and, under
`rocprofv3 --kernel-trace --stats -- python3 tools/per_slice_profile.py`, the kernel time inside it (GPU-busy share).
  python3 tools/per_slice_profile.py [batch=1] [n_slices=16] [overlap=auto|0] [reps=3]
With SMI=1 a child process samples `rocm-smi -P -c` every 0.5 s (sustained clock / power under the load)."""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protosam_amd.runner import build_protosam, support_set, run_slices, part_assign
from protosam_amd.synth import synth_volume

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
overlap = sys.argv[3] if len(sys.argv) > 3 else "auto"
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dev = torch.device("cuda:0")
model, _ = build_protosam(dev, sam_type="vit_h", image_size=512, seed=1234)
vol, lab = synth_volume(64, 512, seed=0, kind="ct")
svol, slab = synth_volume(64, 512, seed=1, kind="ct")
vol_d = vol.to(dev)
sup_imgs, sup_masks = support_set(svol.to(dev), slab.to(dev))
parts = [[z for z in range(64) if part_assign(z, 64) == pt] for pt in range(3)]
for pt in range(3):
    run_slices(model, vol_d, sup_imgs, sup_masks, parts[pt][:1], dev, batch=1)
model.overlap_streams = overlap
zs = parts[1][:n]
for _ in range(2):
    run_slices(model, vol_d, sup_imgs, sup_masks, zs, dev, batch=batch)
torch.cuda.synchronize()
smi = None
if os.environ.get("SMI"):
    smi = subprocess.Popen(["bash", "-c", "while true; do rocm-smi -P -c --csv 2>/dev/null | tail -n +2 | head -2; sleep 0.5; done"],
                           stdout=open(os.environ.get("SMI_OUT", "/tmp/smi.log"), "w"))
t = time.perf_counter()
for _ in range(reps):
    run_slices(model, vol_d, sup_imgs, sup_masks, zs, dev, batch=batch)
torch.cuda.synchronize()
dt = time.perf_counter() - t
if smi:
    smi.terminate()
print(f"batch {batch} overlap {overlap}: {reps * len(zs) / dt:.1f} slices/s, {dt / (reps * len(zs)) * 1e3:.3f} ms per slice, wall {dt * 1e3:.1f} ms for {reps * len(zs)} slices")
