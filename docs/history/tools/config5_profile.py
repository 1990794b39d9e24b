"""BASELINE config 5 (MedSAM ViT-B + 4-class prototype bank, 1024x1024) alone, for rocprofv3 (bench.config5_leg's call, 8 repetitions)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib.util
import torch
from protosam_amd import ops
spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
dev = torch.device("cuda:0")
res = {}
def leg(name, fn, n_units, reps, unit="slices/s", note="", library_default=False):
    fn(); fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(8):
        fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print(f"{name}: {8 * n_units / dt:.1f} {unit}, {dt / 8 * 1e3:.2f} ms per call")
bench.config5_leg(dev, torch, ops, leg)
