#!/usr/bin/env python3
"""Headline benchmark: query-slices/sec (512x512), end-to-end ProtoSAM inference (BASELINE.json).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (config 4 of BASELINE.json, the one the metric is quoted on; it fits one GPU because slices are independent):
DINOv2 ViT-B/14 encoder + ALP prototype match + SAM ViT-H image encoder + prompt encoder + two-way mask decoder on
512x512 slices of a synthetic CT-like volume, seeded random weights, reference default flags (use_bbox, use_points,
point_mode='both', use_cca=False). A "step" is one pass of `ProtoSAM.forward` over a batch of `--batch` query slices
per rank, followed by the all-gather of the step's uint8 masks. Inputs are resident in HBM when the timed region starts.
Support features / prototype banks are cached per z-part (values identical to the reference's per-slice re-encode,
SURVEY Q18); `--no-support-cache` measures the reference's behaviour.

Prints ONE JSON line on rank 0, with `roofline` (dominant kernel = the MFMA GEMM, timed live with HIP events on its own
stream) and `cpu_baseline` (the CPU oracle on a bounded sample of the same workload, timed on this host's cores).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F16_TFLOPS = 2500.0  # dense fp16/bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=16, help="query slices per rank per step")
    ap.add_argument("--micro", type=int, default=16, help="slices pushed through the kernels together (1 = per-slice "
                    "ProtoSAM.forward exactly as the reference caller; >1 = ProtoSAM.forward_batch)")
    ap.add_argument("--sam", default="vit_h", choices=["vit_b", "vit_l", "vit_h"])
    ap.add_argument("--slices", type=int, default=64)
    ap.add_argument("--no-support-cache", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sam-depth", type=int, default=None, help="debug: truncate both models' SAM depth")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    if world != args.gpus and rank == 0:
        log(f"note: --gpus {args.gpus} but WORLD_SIZE {world}; using WORLD_SIZE")
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")

    from protosam_amd import ops
    from protosam_amd.runner import build_protosam, gather_masks, run_slices, support_set
    from protosam_amd.synth import synth_volume

    t0 = time.time()
    model, alp_sd = build_protosam(dev, sam_type=args.sam, image_size=512, seed=1234, sam_depth=args.cpu_sam_depth,
                                   cache_support=not args.no_support_cache)
    vol, lab = synth_volume(args.slices, 512, seed=0, kind="ct")
    svol, slab = synth_volume(args.slices, 512, seed=1, kind="ct")
    vol_d = vol.to(dev)
    sup_imgs, sup_masks = support_set(svol.to(dev), slab.to(dev))
    if rank == 0:
        log(f"built model + volume in {time.time() - t0:.1f}s (world {world})")

    B = args.batch
    out = torch.zeros((B, 512, 512), dtype=torch.uint8, device=dev)

    from protosam_amd.runner import part_assign
    parts = [[z for z in range(args.slices) if part_assign(z, args.slices) == pt] for pt in range(3)]

    def step(s):
        # a step = one window of W*B consecutive slices of one z-part (the caller walks a scan part by part,
        # validation_protosam.py:352-362); rank r takes the slices z = r (mod W) of the window
        pz = parts[s % 3]
        base = (s // 3) * B * world
        zs = [pz[(base + j * world + rank) % len(pz)] for j in range(B)]
        zs.sort()
        masks, st = run_slices(model, vol_d, sup_imgs, sup_masks, zs, dev, out=out, batch=args.micro)
        full = gather_masks(masks, world)
        return zs, full, st

    # setup, not a step: build the three z-parts' support banks and size the workspaces once (the caller walks a scan part
    # by part and the support of a part is constant, validation_protosam.py:355-362), so that neither the W warm-up steps
    # nor the K timed ones depend on which part a step index happens to fall into
    if not args.no_support_cache:
        for pt in range(3):
            run_slices(model, vol_d, sup_imgs, sup_masks, parts[pt][:1], dev, batch=1)
    for s in range(args.warmup):
        step(s)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    timer = ops.KernelTimer() if rank == 0 else None
    ops.GEMM_TIMER = timer
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    ncomp = []
    for s in range(args.steps):
        zs, full, st = step(args.warmup + s)
        ncomp += st
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t1
    ops.GEMM_TIMER = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    n_slices_done = world * B * args.steps
    value = n_slices_done / elapsed
    nl, tg, fl = timer.summary()
    achieved = fl / tg / 1e12 if tg > 0 else 0.0
    # HBM-side bytes per GEMM launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on this same
    # command, gfx950 read correction applied; see profiles/r01_n_gemm_pmc_traffic.json). None if the file is absent.
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_n_gemm_pmc_traffic.json")) as f:
            traffic = json.load(f)["all_gemm_launches"]["bytes_per_launch_avg"]
    except Exception:
        pass
    roofline = {"bound": "mfma", "kernel": "psam_gemm_f16 (gemm8kp_f16_kernel persistent 256x256x64 8-phase / gemm_f16_kernel 128x128x64)",
                "achieved": round(achieved, 2), "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_F16_TFLOPS, 4), "traffic": traffic, "launches": nl,
                "avg_launch_us": round(tg / max(nl, 1) * 1e6, 2), "flop_per_launch_avg": round(fl / max(nl, 1)),
                "gemm_time_share": round(tg / elapsed, 3)}
    cpu = None
    parity = None
    if world == 1 and not args.no_cpu_baseline:
        cpu, parity = cpu_baseline(model, alp_sd, vol, svol, slab, args, dev)
    res = {
        "metric": "query-slices/sec (512x512) end-to-end ProtoSAM infer",
        "value": round(value, 3), "unit": "slices/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": f"ProtoSAM.forward per 512x512 slice: DINOv2 ViT-B/14 + ALP + SAM {args.sam} "
                               f"(encoder + prompt encoder + mask decoder), synthetic CT-like volume",
                   "slices_per_step_per_gpu": B, "micro_batch": args.micro, "volume_slices": args.slices,
                   "support_cached": not args.no_support_cache,
                   "mean_components_per_slice": round(sum(ncomp) / max(len(ncomp), 1), 2),
                   "flags": "use_bbox use_points point_mode=both use_cca=False", "weights": "seeded random (1234)"},
        "roofline": roofline, "cpu_baseline": cpu,
    }
    if parity is not None:
        res["parity_vs_cpu_oracle"] = parity
    print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(model, alp_sd, vol, svol, slab, args, dev):
    """CPU oracle (kind 'port': our pinned restatement of the reference) on a bounded sample: ONE slice of the same
    workload with all host threads. Doubles as a full-depth parity check of the GPU result for that slice."""
    import numpy as np
    from oracle import alp as oalp, dinov2 as odino, glue
    from protosam_amd.runner import part_assign, support_set
    from protosam_amd.synth import synth_state_dict
    # all host cores up to 32: beyond that PyTorch's intra-op parallelism stops scaling on these layer sizes
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    z = args.slices // 2
    sup_imgs, sup_masks = support_set(svol, slab)
    part = part_assign(z, args.slices)
    q = vol[z][None, None].repeat(1, 3, 1, 1).contiguous()
    enc_sd = {k[len("encoder."):]: v for k, v in alp_sd.items() if k.startswith("encoder.")}
    sam_sd = {k: v.detach().cpu().float() for k, v in model.sam.state_dict().items()}
    enc = lambda im: odino.forward_features(im, enc_sd, "dinov2_b14")["x_norm_patchtokens"]  # noqa: E731
    taps = {}
    with torch.no_grad():
        t0 = time.perf_counter()
        logits = oalp.fewshot_forward(enc, sup_imgs[part], sup_masks[part], q, 512)
        pred_ref, scores_ref = glue.protosam_forward(q, logits, sam_sd, args.sam, use_bbox=True, use_points=True,
                                                     point_mode="both", use_cca=False,
                                                     encoder_depth=args.cpu_sam_depth, taps=taps)
        dt = time.perf_counter() - t0
    log(f"cpu_baseline: 1 slice in {dt:.1f}s on {cores} threads")
    # GPU result for the same slice
    from protosam_amd.runner import run_slices
    sup_d = [s.to(dev) for s in sup_imgs]
    msk_d = [m.to(dev) for m in sup_masks]
    masks, _ = run_slices(model, vol.to(dev), sup_d, msk_d, [z], dev)
    g = masks[0].cpu().float()
    r = pred_ref.float()
    tp = (g * r).sum()
    dice = float(2 * tp / (2 * tp + ((1 - g) * r).sum() + (g * (1 - r)).sum() + 1e-8))
    st = model.last_stats
    parity = {"slice": z, "dice_final_mask": round(dice, 5), "flipped_pixels": int((g != r).sum()),
              "components": int(st.get("n_prompts", 0))}
    if "low_res" in st and len(taps.get("low_res", [])) == st["low_res"].shape[0]:
        low = st["low_res"][:, st["sel"]].cpu()
        low_ref = torch.stack([l[0] for l in taps["low_res"]])
        parity["max_abs_dprob_low_res"] = float((torch.sigmoid(low) - torch.sigmoid(low_ref)).abs().max())
        parity["max_abs_dscore"] = float(np.abs(np.array([float(v) for v in st["iou"][:, st["sel"]].cpu()]) -
                                                np.array(scores_ref)).max())
    cpu = {"value": round(1.0 / dt, 5), "unit": "slices/s", "cores": cores, "kind": "port",
           "sample": f"1 slice (z={z}) of the same volume through the full CPU oracle pipeline (fp32, {cores} threads), "
                     f"{dt:.1f} s"}
    return cpu, parity


if __name__ == "__main__":
    main()
