"""Per-slice `ProtoSAM.forward` (micro_batch 1): wall time per slice next to the GPU-side sum of kernel times per stage
(ops.TIMERS / GEMM_TIMER events), to see whether the reference-shaped call pattern is launch- or kernel-bound.
Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protosam_amd import ops
from protosam_amd.runner import build_protosam, run_slices, support_set
from protosam_amd.synth import synth_volume

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
model, _ = build_protosam(dev, sam_type="vit_h", image_size=512, seed=1234)
vol, _ = synth_volume(64, 512, seed=0, kind="ct")
svol, slab = synth_volume(64, 512, seed=1, kind="ct")
sup_imgs, sup_masks = support_set(svol.to(dev), slab.to(dev))
vol_d = vol.to(dev)
zs = list(range(20, 20 + n))
run_slices(model, vol_d, sup_imgs, sup_masks, zs[:2], dev, batch=1)
torch.cuda.synchronize()
ops.GEMM_TIMER = ops.KernelTimer()
t0 = time.perf_counter()
run_slices(model, vol_d, sup_imgs, sup_masks, zs, dev, batch=1)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
nl, tg, fl = ops.GEMM_TIMER.summary()
ops.GEMM_TIMER = None
print(f"per-slice forward: {dt / n * 1e3:.2f} ms/slice wall ({n / dt:.1f} slices/s); GEMM {tg / n * 1e3:.2f} ms/slice over "
      f"{nl // n} launches ({fl / tg / 1e12:.0f} TFLOP/s)")
t0 = time.perf_counter()
run_slices(model, vol_d, sup_imgs, sup_masks, zs, dev, batch=1)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"without the GEMM event pairs: {dt / n * 1e3:.2f} ms/slice wall ({n / dt:.1f} slices/s)")
