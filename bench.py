#!/usr/bin/env python3
"""Headline benchmark: query-slices/sec (512x512), end-to-end ProtoSAM inference (BASELINE.json).

  python bench.py --gpus N --steps K --warmup W
      N > 1 without a launcher: this process stays off the GPU and starts N fresh ranks itself
      (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py ...),
      relays rank 0's JSON line and exits with the ranks' status. Under a launcher (WORLD_SIZE in the
      environment) it is one rank of that job.

Workload (config 4 of BASELINE.json, the one the metric is quoted on; it fits one GPU because slices are independent):
DINOv2 ViT-B/14 encoder + ALP prototype match + SAM ViT-H image encoder + prompt encoder + two-way mask decoder on
512x512 slices of a synthetic CT-like volume, seeded random weights, reference default flags (use_bbox, use_points,
point_mode='both', use_cca=False). A "step" is one pass of `ProtoSAM.forward` over a batch of `--batch` query slices
(default 32) per rank, followed by the all-gather of the step's uint8 masks. Inputs are resident in HBM when the timed region starts.
Support features / prototype banks are cached per z-part (values identical to the reference's per-slice re-encode,
SURVEY Q18); the line also carries the reference-shaped numbers: `per_slice_forward` (one `ProtoSAM.forward` per slice, what
validation_protosam.py:387 does) and `no_support_cache` (support re-encoded for every slice, grid_proto_fewshot.py:181-184).

Prints ONE JSON line on rank 0, with `roofline` (dominant kernel = the MFMA GEMM, timed live with HIP events on its own
stream), `roofline_hbm` (the HBM-bound kernels: achieved algorithmic GB/s), per-stage times, and `cpu_baseline` (the CPU
oracle on a bounded sample of the same workload, timed on this host's cores at all cores and at 1 thread).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F16_TFLOPS = 2500.0  # dense fp16/bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
# What the chip SUSTAINS on v_mfma_f32_32x32x16_f16 alone, from registers, on all 256 CUs with random operands: the socket power cap
# pulls the shader clock from 2.39 to 1.61 GHz (zero operands: 2486 TFLOP/s at 2.39 GHz). tools/micro/mfma_power_bench.hip,
# profiles/r04_mfma_power_limit.txt. Reported beside `peak`, never instead of it.
SUSTAINED_F16_TFLOPS = 1624.0
PEAK_HBM_GBS = 8000.0     # HBM3E peak (6.3 TB/s achievable by a float4 copy), same guide
PEAK_F32_MFMA_TFLOPS = 157.0   # v_mfma_f32_32x32x2_f32, same guide (tools/micro/mfma_f32_rate.hip measures 149-155)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    # (round 6: 32 slices per step instead of 16 - the same kernels on twice the rows: fewer launches and tile-list tails per slice,
    # +2.5 ... 3.5 % at 32 / 48 / 64 slices per call, measured on one box; a step's batch then spans two z-parts of the volume, which
    # forward_batch takes as a mixed-support batch)
    ap.add_argument("--batch", type=int, default=32, help="query slices per rank per step")
    ap.add_argument("--micro", type=int, default=32, help="slices pushed through the kernels together (1 = per-slice "
                    "ProtoSAM.forward exactly as the reference caller; >1 = ProtoSAM.forward_batch)")
    ap.add_argument("--sam", default="vit_h", choices=["vit_b", "vit_l", "vit_h"])
    ap.add_argument("--slices", type=int, default=64)
    ap.add_argument("--no-support-cache", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the per-slice / no-cache / stage / HBM-roofline legs")
    ap.add_argument("--cpu-slices", type=int, default=3, help="slices of the CPU baseline at all cores")
    ap.add_argument("--cpu-1thread-slices", type=int, default=1, help="slices of the CPU baseline at 1 thread, as the reference "
                    "runs (validation_protosam.py:299: one slice takes about a minute); 0 skips it")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"], help="weak: --batch slices per rank per step; strong: "
                    "a step is ONE pass over the --slices volume, rank r takes z = r (mod world) (SURVEY 8e), one all-gather per step")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the BASELINE configs 2 / 3 / 5 legs of the default run")
    ap.add_argument("--cpu-sam-depth", type=int, default=None, help="debug: truncate both models' SAM depth")
    return ap.parse_args(argv)


def step_slices(s, parts, B, world, rank):
    """Slice indices rank `rank` processes in step `s`: a step = one window of world*B consecutive slices of one z-part (the
    caller walks a scan part by part, validation_protosam.py:352-362) - or, when the window is larger than a part, of the volume;
    rank r takes z = r (mod world) of the window. The union over ranks does not depend on `world` for a given world*B."""
    if B * world > min(len(p) for p in parts):
        # a window larger than a z-part: world*B consecutive slices of the VOLUME (the batch spans parts: a mixed-support batch)
        n = sum(len(p) for p in parts)
        base = s * B * world
        return sorted((base + j * world + rank) % n for j in range(B))
    pz = parts[s % 3]
    base = (s // 3) * B * world
    zs = [pz[(base + j * world + rank) % len(pz)] for j in range(B)]
    zs.sort()
    return zs


def strong_slices(n_slices, world, rank):
    """--scaling strong: the slices of the whole volume rank `rank` owns, z = rank (mod world) (protosam_amd.runner.shard_slices)."""
    return list(range(rank, n_slices, world))


def launch_ranks(args):
    """No launcher, --gpus N > 1: start N fresh rank processes. This parent never initialises the GPU."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("launching:", " ".join(cmd))
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    if lines:
        print(lines[-1], flush=True)
    elif p.returncode == 0:
        log("no JSON line from rank 0")
        return 1
    return p.returncode


def _gemm_traffic(args, B):
    """Average measured L2<->fabric bytes per launch of the four SAM block GEMMs (97 % of the GEMM FLOPs of a step) from the newest
    profiles/r*_gemm_traffic_by_shape.json - or None when that file was measured on other kernel sources (git blob hashes differ),
    on another batch shape, or is missing."""
    import glob
    import hashlib
    import json as _json
    root = os.path.dirname(os.path.abspath(__file__))
    files = sorted(glob.glob(os.path.join(root, "profiles", "r*_gemm_traffic_by_shape.json")))
    if not files or args.sam != "vit_h":
        return None
    try:
        doc = _json.load(open(files[-1]))
        for path, h in doc["sources"].items():
            data = open(os.path.join(root, path), "rb").read()
            if hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest() != h:
                return None
        M = min(args.micro, B) * 4096
        tot = 0
        for key in (f"{M}x3840x1280x0", f"{M}x1280x1280x2", f"{M}x5120x1280x1", f"{M}x1280x5120x2"):
            sh = doc["shapes"][key]
            tot += sh["read_bytes_per_launch"] + sh["write_bytes_per_launch"]
        return {"bytes_per_launch_avg": tot // 4, "file": os.path.relpath(files[-1], root),
                "note": "mean over the qkv / proj / fc1 / fc2 launches of a SAM ViT-H block at this batch; L2<->fabric requests "
                        "(Infinity-Cache hits included: upper bound of HBM traffic), FETCH_SIZE doubled per the gfx950 correction"}
    except Exception:
        return None


class PowerSampler:
    """Socket power and shader clock of this rank's GPU from the amdgpu hwmon files (power1_input in uW, freq1_input in Hz), sampled by
    a thread every 20 ms between start() and stop(): evidence for the power cap that binds the fp16 MFMA rate. Absent
    files (another driver, no permission) => None, nothing else changes."""

    def __init__(self, index):
        import glob
        cards = [c for c in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")
                 if os.path.exists(os.path.join(c, "power1_input")) and os.path.exists(os.path.join(c, "freq1_input"))]
        self.dir = None
        try:        # the card whose PCI address is this rank's device (a box can expose the hwmon files of GPUs it does not own)
            import torch
            pr = torch.cuda.get_device_properties(index)
            bdf = "%04x:%02x:%02x." % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            for c in cards:
                if os.path.basename(os.path.realpath(os.path.join(c, "..", ".."))).startswith(bdf):
                    self.dir = c
        except Exception:
            pass
        if self.dir is None and len(cards) == 1:
            self.dir = cards[0]
        self.samples, self.run, self.thread = [], False, None

    def _read(self, name):
        with open(os.path.join(self.dir, name)) as f:
            return float(f.read().strip())

    def _loop(self):
        while self.run:
            try:
                self.samples.append((self._read("power1_input") * 1e-6, self._read("freq1_input") * 1e-6))
            except Exception:
                return
            time.sleep(0.02)

    def start(self):
        if self.dir is None:
            return
        import threading
        self.samples, self.run = [], True
        self.thread = threading.Thread(target=self._loop, daemon=True)
        self.thread.start()

    def stop(self):
        if self.thread is None:
            return None
        self.run = False
        self.thread.join()
        self.thread = None
        if len(self.samples) < 3:
            return None
        pw = [a for a, _ in self.samples]; fq = [b for _, b in self.samples]
        try:
            cap = self._read("power1_cap") * 1e-6
        except Exception:
            cap = None
        return {"avg_w": round(sum(pw) / len(pw), 1), "max_w": round(max(pw), 1), "cap_w": cap, "sclk_mhz_avg": round(sum(fq) / len(fq)),
                "sclk_mhz_min": round(min(fq)), "sclk_mhz_max": round(max(fq)), "samples": len(pw),
                "source": "amdgpu hwmon power1_input / freq1_input, 20 ms period, over the timed steps"}


class DeviceRuntime:
    """What a rank's body needs from the machine. This one is the product's: an MI355X per rank, RCCL (`nccl`) between them, HIP events
    on the launch stream. tests/test_runner_dist_cpu.py substitutes a host runtime (gloo, stubbed `run_slices`) so that the N > 1 body -
    argument handling, step indexing, the all-gather, the reductions, the JSON line, the teardown - runs at world 2 without a GPU."""
    backend = "nccl"

    def __init__(self):
        # PSAM_BENCH_BACKEND=gloo + PSAM_BENCH_SHARE_GPU=1: the whole N > 1 body with live models on a box with ONE GPU (every rank on
        # cuda:0, collectives staged through the host) - a functional check of the multi-rank path, never a throughput number
        self.backend = os.environ.get("PSAM_BENCH_BACKEND", "nccl")
        self.share_gpu = os.environ.get("PSAM_BENCH_SHARE_GPU", "0") != "0"

    def device(self, local_rank):
        import torch
        idx = 0 if self.share_gpu else local_rank
        torch.cuda.set_device(idx)
        return torch.device(f"cuda:{idx}")

    def init_group(self, local_rank):
        import torch
        import torch.distributed as dist
        if self.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{0 if self.share_gpu else local_rank}"))
        else:
            dist.init_process_group(self.backend)

    def sync(self):
        import torch
        torch.cuda.synchronize()

    def event(self):
        import torch
        return torch.cuda.Event(enable_timing=True)

    def kernel_timer(self):
        from protosam_amd import ops
        return ops.KernelTimer()

    def power_sampler(self, local_rank):
        return PowerSampler(local_rank)

    def build(self, args, dev):
        """-> (model, alp_sd, vol, svol, slab, vol_d, sup_imgs, sup_masks): seeded weights and volumes, resident on `dev`"""
        from protosam_amd.runner import build_protosam, support_set
        from protosam_amd.synth import synth_volume
        model, alp_sd = build_protosam(dev, sam_type=args.sam, image_size=512, seed=1234, sam_depth=args.cpu_sam_depth,
                                       cache_support=not args.no_support_cache)
        vol, lab = synth_volume(args.slices, 512, seed=0, kind="ct")
        svol, slab = synth_volume(args.slices, 512, seed=1, kind="ct")
        sup_imgs, sup_masks = support_set(svol.to(dev), slab.to(dev))
        return model, alp_sd, vol, svol, slab, vol.to(dev), sup_imgs, sup_masks

    def run_slices(self, *a, **k):
        from protosam_amd.runner import run_slices
        return run_slices(*a, **k)


def main(argv=None, runtime=None):
    args = parse_args(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))
    rt = runtime or DeviceRuntime()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        rt.init_group(local_rank)
        # the ranks of one node share its host cores: the prompt logic between the kernels is single-threaded, and a rank that spins
        # os.cpu_count() intra-op threads for the few host-side torch ops would only steal them from its neighbours
        torch.set_num_threads(max(1, (os.cpu_count() or 1) // world))
    if world != args.gpus and rank == 0:
        log(f"note: --gpus {args.gpus} but WORLD_SIZE {world}; using WORLD_SIZE")
    dev = rt.device(local_rank)

    from protosam_amd import ops, protosam as psmod
    from protosam_amd.runner import all_gather_rows, gather_masks, part_assign

    t0 = time.time()
    model, alp_sd, vol, svol, slab, vol_d, sup_imgs, sup_masks = rt.build(args, dev)
    if rank == 0:
        log(f"built model + volume in {time.time() - t0:.1f}s (world {world})")
    ranks_seen = [0]
    if world > 1:
        # every rank's id through the same collective the masks take (RCCL all-gather on the device): the job really has `world`
        # distinct ranks talking to each other before anything is timed
        ids = all_gather_rows(torch.tensor([rank], dtype=torch.int32, device=dev))
        ranks_seen = sorted(int(v) for v in ids.cpu())
        if ranks_seen != list(range(world)):
            raise SystemExit(f"rank {rank}: the all-gather of rank ids returned {ranks_seen}, expected 0..{world - 1}")

    # The headline is measured with the two encoders on ONE stream: `roofline` prices every GEMM launch with its own pair of HIP events
    # (and must agree with rocprofv3's per-kernel durations), which concurrent kernels of a second stream would blur. The library's
    # default ("auto": SAM encoder beside DINOv2 + ALP on a second stream once the calls are dense) is reported as a leg of its own.
    if model is not None:
        model.overlap_streams = "0"
    strong = args.scaling == "strong"
    if strong and args.slices % world:
        raise SystemExit("--scaling strong needs --slices divisible by the number of ranks")
    B = args.slices // world if strong else args.batch
    out = torch.zeros((B, 512, 512), dtype=torch.uint8, device=dev)
    parts = [[z for z in range(args.slices) if part_assign(z, args.slices) == pt] for pt in range(3)]
    ag_events = []                                # (start, end) HIP events around the all-gather of a timed step (world > 1)

    def step(s, micro=None, volume=None, timed=False):
        zs = strong_slices(args.slices, world, rank) if strong else step_slices(s, parts, B, world, rank)
        masks, st = rt.run_slices(model, vol_d if volume is None else volume, sup_imgs, sup_masks, zs, dev, out=out,
                                  batch=micro or args.micro)
        if timed and world > 1:
            e0, e1 = rt.event(), rt.event()
            e0.record()
            full = gather_masks(masks, world)
            e1.record()
            ag_events.append((e0, e1))
        else:
            full = gather_masks(masks, world)
        return zs, full, st

    # setup, not a step: build the three z-parts' support banks and size the workspaces once (the caller walks a scan part
    # by part and the support of a part is constant, validation_protosam.py:355-362), so that neither the W warm-up steps
    # nor the K timed ones depend on which part a step index happens to fall into
    if not args.no_support_cache:
        for pt in range(3):
            rt.run_slices(model, vol_d, sup_imgs, sup_masks, parts[pt][:1], dev, batch=1)
    for s in range(args.warmup):
        step(s)
    rt.sync()
    if world > 1:
        dist.barrier()
    # The GEMM launches are priced with their own pair of HIP events on every THIRD timed step (all steps launch the same shapes): a
    # pair per launch on every step cost the headline 1-2 % (3600 launches in six steps; 155.2 against 158.2 slices/s in the same
    # minute on one box), and `value` is the job's throughput, not the instrument's. `roofline.launches` counts the priced launches.
    timer = rt.kernel_timer() if rank == 0 else None
    timed_steps = [s for s in range(args.steps) if s % 3 == 0]
    step_ev = [rt.event() for _ in range(args.steps + 1)]
    rt.sync()
    t1 = time.perf_counter()
    step_ev[0].record()
    power = rt.power_sampler(local_rank) if rank == 0 else None
    if power:
        power.start()
    ncomp = []
    full = None
    for s in range(args.steps):
        ops.GEMM_TIMER = timer if s in timed_steps else None
        zs, full, st = step(args.warmup + s, timed=True)
        step_ev[s + 1].record()
        ncomp += st
    rt.sync()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t1
    power_clock = power.stop() if power else None
    ops.GEMM_TIMER = None
    per_rank_ms, allgather_ms = None, None
    if world > 1:
        # per-rank wall time of the timed region (its max is the job's time) and each rank's mean all-gather time, on rank 0's line
        mine = torch.tensor([elapsed, sum(a.elapsed_time(b) for a, b in ag_events) / max(len(ag_events), 1) * 1e-3],
                            dtype=torch.float64, device=dev)
        every = all_gather_rows(mine.view(1, 2)).cpu()
        per_rank_ms = [round(float(v) / args.steps * 1e3, 3) for v in every[:, 0]]
        allgather_ms = [round(float(v) * 1e3, 3) for v in every[:, 1]]
        elapsed = float(every[:, 0].max())

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return {"rank": rank, "last_zs": zs, "last_gather": full}

    import numpy as np
    step_ms = [step_ev[i].elapsed_time(step_ev[i + 1]) for i in range(args.steps)]
    n_slices_done = world * B * args.steps
    value = n_slices_done / elapsed
    nl, tg, fl = timer.summary()
    achieved = fl / tg / 1e12 if tg > 0 else 0.0
    roofline = {"bound": "mfma", "kernel": "psam_gemm_f16 (fp16 operands, fp32 accumulate; every Linear / conv-as-GEMM of the "
                                           "two ViT encoders; the launches also carry the blocks' LayerNorms, folded into "
                                           "their epilogues - no separate LayerNorm pass runs)",
                "achieved": round(achieved, 2), "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_F16_TFLOPS, 4),
                # HBM-side bytes per launch come from rocprofv3 PMC passes, which cannot run inside this process: taken from the
                # per-shape file of tools/gemm_traffic_by_shape.sh ONLY when it was measured on the shipped kernel sources
                "traffic": (_gemm_traffic(args, B) or {}).get("bytes_per_launch_avg"), "traffic_detail": _gemm_traffic(args, B), "launches": nl,
                "avg_launch_us": round(tg / max(nl, 1) * 1e6, 2), "flop_per_launch_avg": round(fl / max(nl, 1)),
                "priced_steps": len(timed_steps), "gemm_time_share": round(tg * args.steps / max(len(timed_steps), 1) / elapsed, 3),
                "sustained_mfma_only": {"value": SUSTAINED_F16_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / SUSTAINED_F16_TFLOPS, 4),
                                        "note": "register-only MFMA loop, random fp16 operands, 256 CUs: power-capped at 1.61 GHz "
                                                "(profiles/r04_mfma_power_limit.txt); `peak` is the nominal 2.39 GHz figure"}}
    # the GEMM launches of the timed steps by shape: the four Linear layers of a SAM block as the pipeline runs them (bias / GELU /
    # residual + LayerNorm epilogues included), each against the MFMA peak AND against HBM on its algorithmic bytes - the proj shape
    # (K = N = 1280 with an fp32 residual: 210 FLOP per byte, below the machine balance of 2500 / 8 = 312) is bound by bytes, not by MFMAs
    by_shape = []
    for (Mg, Ng, Kg, epi, folded), (cnt, secs, flops) in sorted(timer.by_tag().items(), key=lambda kv: -kv[1][1])[:8]:
        byts = 2.0 * Mg * Kg + 2.0 * Ng * Kg + (8.0 * Mg * Ng + (2.0 * Mg * Ng + 8.0 * Mg * (Ng // 64) if folded else 0.0) if epi == 2 else 2.0 * Mg * Ng)
        by_shape.append({"M": Mg, "N": Ng, "K": Kg, "epilogue": ["fp16", "gelu fp16", "fp32 residual", "relu fp16"][epi] + (" + folded LayerNorm" if folded else ""),
                         "launches": cnt, "avg_us": round(secs / cnt * 1e6, 1),
                         "share_of_step": round(secs * args.steps / max(len(timed_steps), 1) / elapsed, 4),
                         "tflops": round(flops / secs / 1e12, 1), "mfma_frac": round(flops / secs / 1e12 / PEAK_F16_TFLOPS, 4),
                         "algorithmic_gb_per_launch": round(byts / 1e9, 3), "hbm_gbs": round(byts * cnt / secs / 1e9, 1),
                         "hbm_frac": round(byts * cnt / secs / 1e9 / PEAK_HBM_GBS, 4),
                         "bound": "hbm" if flops / cnt / byts < PEAK_F16_TFLOPS * 1e3 / PEAK_HBM_GBS else "mfma"})
    roofline["by_shape"] = by_shape
    res = {
        "metric": "query-slices/sec (512x512) end-to-end ProtoSAM infer",
        "value": round(value, 3), "unit": "slices/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        # N > 1: which ranks answered the RCCL all-gather of rank ids, every rank's own wall time per step (`ms_per_step` is their
        # maximum) and its mean time inside the step's all-gather of uint8 masks (HIP events around the collective: it includes
        # waiting for the slowest rank's masks, so it is an upper bound of the transfer itself)
        "ranks_seen": ranks_seen, "per_rank_ms_per_step": per_rank_ms, "allgather_ms_per_step": allgather_ms,
        "collective_backend": (dist.get_backend() if world > 1 else None),
        "host_threads_per_rank": torch.get_num_threads(),
        "ms_per_step_std": round(float(np.std(step_ms)), 3), "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": f"ProtoSAM.forward per 512x512 slice: DINOv2 ViT-B/14 + ALP + SAM {args.sam} "
                               f"(encoder + prompt encoder + mask decoder), synthetic CT-like volume",
                   "slices_per_step_per_gpu": B, "micro_batch": args.micro, "volume_slices": args.slices,
                   "support_cached": not args.no_support_cache,
                   "mean_components_per_slice": round(sum(ncomp) / max(len(ncomp), 1), 2),
                   "flags": "use_bbox use_points point_mode=both use_cca=False", "weights": "seeded random (1234)"},
        "roofline": roofline,
        "power_clock": power_clock,
    }
    if world == 1 and not args.no_extras:     # single-GPU legs (they would need the other ranks for the all-gather otherwise)
        res.update(extras(args, model, step, ops, psmod, B, torch, dev))
        if not strong:
            res["rank_of_8_strong"] = rank_of_8(args, model, vol_d, sup_imgs, sup_masks, dev, torch)
        if not args.no_other_configs:
            res["other_configs"] = other_configs(args, dev, torch, ops)
    cpu = parity = None
    if world == 1 and not args.no_cpu_baseline:
        # (rank 0 at N = 1 only, after every timed region of the run)
        cpu, parity = cpu_baseline(model, alp_sd, vol, svol, slab, args, dev)
    res["cpu_baseline"] = cpu
    if parity is not None:
        res["parity_vs_cpu_oracle"] = parity
    print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return {"rank": rank, "last_zs": zs, "last_gather": full, "line": res}


def extras(args, model, step, ops, psmod, B, torch, dev):
    """Single-rank legs after the headline measurement (rank 0 only, no collectives): the reference-shaped numbers, per-stage
    GPU time and the HBM-side roofline entries. Each leg is a couple of steps."""
    out = {}
    s0 = args.warmup + args.steps

    def timed(n, **kw):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for i in range(n):
            step(s0 + i, **kw)
        torch.cuda.synchronize()
        return time.perf_counter() - t
    # (a) stage times + HBM-side kernels, on the headline configuration (2 steps with the timers on)
    names = ("layernorm", "attention_window", "attention_global", "alp_sim", "prob_argmax", "ccl")
    for n in names:
        ops.TIMERS[n] = ops.KernelTimer()
    psmod.STAGE_TIMER = psmod.StageTimer()
    nst = 2
    timed(nst)
    stages = psmod.STAGE_TIMER.summary()
    psmod.STAGE_TIMER = None
    out["stage_ms_per_step"] = {k: round(v / nst, 3) for k, v in stages.items()}
    hbm = []
    for n in names:
        nl, tt, by = ops.TIMERS[n].summary()
        if nl == 0 or tt <= 0:
            continue
        e = {"kernel": n, "bound": "hbm", "achieved": round(by / tt / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
             "frac": round(by / tt / 1e9 / PEAK_HBM_GBS, 4), "launches_per_step": nl // nst,
             "avg_launch_us": round(tt / nl * 1e6, 2), "bytes_per_launch": round(by / nl)}
        fl = ops.TIMERS[n].summary2()
        if fl > 0 and n == "alp_sim":   # fp32 operands: priced against the fp32-MFMA rate (its real roof; FLOPs at bank capacity: upper bound)
            e["mfma_tflops"] = round(fl / tt / 1e12, 1)
            e["mfma_frac"] = round(fl / tt / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)
            e["mfma_peak"] = PEAK_F32_MFMA_TFLOPS
            e["bound"] = "mfma (fp32)"
        elif fl > 0:   # the attention kernels are also priced against the MFMA roofline
            e["mfma_tflops"] = round(fl / tt / 1e12, 1)
            e["mfma_frac"] = round(fl / tt / 1e12 / PEAK_F16_TFLOPS, 4)
        hbm.append(e)
    ops.TIMERS.clear()
    out["roofline_hbm"] = hbm
    # (a2) the library's default stream mode: SAM image encoder on a second stream beside DINOv2 + ALP + connected components.
    #      Interleaved A/B against the single-stream mode in this process: five blocks of two steps each way (round 4 compared a
    #      3-step sample with the headline of another minute of the run: 143 vs 150 on the driver's box, 149 vs 147 the round before)
    import numpy as np
    model.overlap_streams = "auto"
    timed(4)                                                     # ("auto" overlaps after four dense batched calls)
    ab = {"auto": [], "0": []}
    for blk in range(5):
        for mode in ("auto", "0"):
            model.overlap_streams = mode
            ab[mode].append(2 * B / timed(2))
    model.overlap_streams = "0"
    out["overlap_streams_auto"] = {"value": round(float(np.mean(ab["auto"])), 2), "unit": "slices/s", "std": round(float(np.std(ab["auto"])), 2),
                                   "single_stream_same_minute": round(float(np.mean(ab["0"])), 2), "single_stream_std": round(float(np.std(ab["0"])), 2),
                                   "steps_each": 10,
                                   "note": "headline configuration with PSAM_OVERLAP_STREAMS=auto (the library default): the SAM image "
                                           "encoder runs on a second HIP stream beside the coarse model; five interleaved blocks of two "
                                           "steps per mode; per-kernel event timing is blurred by the concurrency, so the headline and "
                                           "its roofline are measured without it"}
    # (b) the reference-shaped call pattern: one ProtoSAM.forward per slice (validation_protosam.py:387)
    model.overlap_streams = "auto"                                # (the library default; no per-kernel timing in this leg)
    timed(1, micro=1)
    dt = timed(1, micro=1)
    model.overlap_streams = "0"
    out["per_slice_forward"] = {"value": round(B / dt, 2), "unit": "slices/s",
                                "note": "micro_batch 1: one ProtoSAM.forward call per slice, support cached, PSAM_OVERLAP_STREAMS=auto "
                                        "(the SAM encoder on a second stream beside the coarse model)"}
    # (b2) a scan whose organ covers only part of the z range and comes with satellites: empty coarse masks (SAM skipped for
    # the slice) and several prompt sets per slice inside the timed region; every step visits all three z-parts
    from protosam_amd.synth import synth_volume
    sparse = synth_volume(args.slices, 512, seed=0, kind="ct_sparse")[0].to(dev)
    nsp = 3
    for i in range(nsp):
        step(s0 + i, volume=sparse)
    torch.cuda.synchronize()
    t = time.perf_counter()
    counts = []
    for i in range(nsp):
        counts += step(s0 + i, volume=sparse)[2]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    out["sparse_volume"] = {"value": round(nsp * B / dt, 2), "unit": "slices/s",
                            "empty_slices": round(sum(1 for c in counts if c == 0) / max(len(counts), 1), 3),
                            "mean_prompt_sets_per_nonempty_slice": round(sum(counts) / max(sum(1 for c in counts if c), 1), 2),
                            "note": "headline configuration on a volume whose organ (with five satellite blobs) spans 44 % of "
                                    "the slices and whose outer 28 % are air: empty coarse masks skip SAM, organ slices carry "
                                    "4-6 prompt sets"}
    # (b3) the SAM image encoder's Linear layers at the REFERENCE'S arithmetic width (ImageEncoderViT.gemm_x3: fp32 operands and results,
    # three fp16 MFMA products on (hi, lo) halves per product; LayerNorm / GELU as fp32 passes; neck and patch embedding on their split
    # forms; the attention products keep fp16 operands): what the fp16-operand headline buys, with its own parity figure
    # (`parity_vs_cpu_oracle.reference_width`, same slices)
    enc = model.sam.image_encoder
    enc.gemm_x3 = True
    try:
        timed(1)
        dt = timed(2)
        out["reference_width"] = {"value": round(2 * B / dt, 2), "unit": "slices/s", "dtype": "f32 Linear layers (3 x f16 MFMA on hi / lo halves), f16 attention operands",
                                  "note": "headline configuration with PSAM_ENCODER_X3=1: every Linear of the SAM ViT blocks through "
                                          "psam_gemm_f32x3 (fp32 in, fp32 out), fp32 LayerNorm and GELU passes, split-fp16 neck and "
                                          "exact-pixel patch embedding; DINOv2 and the decoder as in the headline (the decoder's image side "
                                          "is already at fp32 accuracy). Not a tuned path: the x3 kernel is the decoder's 64 x 128-tile one"}
    finally:
        enc.gemm_x3 = False
    # (c) support re-encoded for every slice as the reference does (grid_proto_fewshot.py:181-184)
    alp = model.coarse_segmentation_model.model
    if alp.cache_support:
        alp.cache_support = False
        alp._sup_cache = []
        try:
            timed(1)
            dt = timed(2)
            out["no_support_cache"] = {"value": round(2 * B / dt, 2), "unit": "slices/s",
                                       "note": "headline configuration with the support image re-encoded for every slice"}
        finally:
            alp.cache_support = True
    return out


def rank_of_8(args, model, vol_d, sup_imgs, sup_masks, dev, torch):
    """What ONE rank of an 8-rank strong-scaling job over the 64-slice volume runs per step (z = 0 mod 8: eight slices, batched per
    z-part as the volume runner does), measured here on one GPU: the per-rank compute efficiency at N = 8 before hardware is."""
    from protosam_amd.runner import run_slices
    zs = strong_slices(args.slices, 8, 0)
    run_slices(model, vol_d, sup_imgs, sup_masks, zs, dev, batch=len(zs))
    torch.cuda.synchronize()
    t = time.perf_counter()
    n = 3
    for _ in range(n):
        run_slices(model, vol_d, sup_imgs, sup_masks, zs, dev, batch=len(zs))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    return {"slices_per_step": len(zs), "ms_per_step": round(dt * 1e3, 2), "slices_per_s_per_rank": round(len(zs) / dt, 2),
            "note": "a rank's 8 slices span the three z-parts (different support sets) and run as ONE mixed-support batch (round 4: both "
                    "encoders, CCL, SAM and the decoder do not depend on the support; the prototype match runs per support set); x8 ranks = "
                    "the compute-side ceiling of the 8-GPU strong-scaling number (the all-gather of 2 MiB of masks per rank is not in it)"}


def other_configs(args, dev, torch, ops):
    """BASELINE.json configs 2 / 3 / 5 as short legs (config 4 is the headline): slices/s and the GEMM rate of each."""
    from protosam_amd.runner import build_protosam, part_assign, run_slices, support_set
    from protosam_amd.synth import synth_volume
    out = {}

    def leg(name, fn, n_units, reps, unit="slices/s", note="", library_default=False):
        fn()
        torch.cuda.synchronize()
        value_dt = None
        if library_default:
            # one-slice-per-call legs: `value` is the library as a caller gets it (HIP-graph replay of the encoder forwards, second
            # stream) - both switch themselves off while per-launch events are attached, so the roofline pass below runs separately
            fn()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            value_dt = time.perf_counter() - t
        timer = ops.KernelTimer()
        ops.GEMM_TIMER = timer
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        ops.GEMM_TIMER = None
        nl, tg, fl = timer.summary()
        ach = fl / tg / 1e12 if tg > 0 else None
        vdt = value_dt if value_dt is not None else dt
        out[name] = {"value": round(reps * n_units / vdt, 2), "unit": unit, "ms_per_call": round(vdt / reps * 1e3, 2),
                     # same definition as the headline's `roofline` (dominant kernel = psam_gemm_f16; one stream: the per-launch events
                     # of this leg are not blurred by a second stream - ProtoSAM's "auto" overlap backs off while a timer is attached)
                     "roofline": {"bound": "mfma", "achieved": round(ach, 1) if ach else None, "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
                                  "frac": round(ach / PEAK_F16_TFLOPS, 4) if ach else None, "launches": nl,
                                  "avg_launch_us": round(tg / max(nl, 1) * 1e6, 2), "gemm_time_share": round(tg / dt, 3)},
                     "note": note}
    # config 2: DINOv2 ViT-B/14 + ALP cosine-similarity map, 512x512, batch 1 (coarse prediction only)
    m2, _ = build_protosam(dev, sam_type="vit_b", image_size=512, seed=1234, sam_depth=1, coarse_pred_only=True)
    vol, _ = synth_volume(32, 512, seed=0, kind="mri")
    svol, slab = synth_volume(32, 512, seed=1, kind="mri")
    vol_d = vol.to(dev)
    sup_imgs, sup_masks = support_set(svol.to(dev), slab.to(dev))
    zs_part = [z for z in range(32) if part_assign(z, 32) == 1]
    leg("config2", lambda: run_slices(m2, vol_d, sup_imgs, sup_masks, zs_part[:8], dev, batch=1), 8, 3, library_default=True,
        note="DINOv2 ViT-B/14 + ALP prototype match only (coarse_pred_only), one ProtoSAM.forward per 512x512 slice, support cached; "
             "value without per-launch events (HIP-graph replay of the encoder forward), the roofline object from a second pass with them")
    leg("config2_batched", lambda: run_slices(m2, vol_d, sup_imgs, sup_masks, zs_part[:8], dev, batch=8), 8, 3,
        note="the same through forward_batch, 8 slices together")
    del m2
    # config 3: + SAM ViT-B mask decoder, 512x512 MRI-like volume of 32 slices
    m3, _ = build_protosam(dev, sam_type="vit_b", image_size=512, seed=1234)
    zs = list(range(32))
    leg("config3", lambda: run_slices(m3, vol_d, sup_imgs, sup_masks, zs, dev, batch=16), 32, 2,
        note="full ProtoSAM with SAM ViT-B on the 32-slice MRI-like volume, two 16-slice batches (a batch spans z-parts: one encoder "
             "forward, the prototype match per support set)")
    leg("config3_per_slice", lambda: run_slices(m3, vol_d, sup_imgs, sup_masks, zs[:16], dev, batch=1), 16, 2, library_default=True,
        note="one ProtoSAM.forward per slice; value without per-launch events (HIP-graph replay of both encoder forwards, second stream), "
             "the roofline object from a second pass with them")
    del m3
    torch.cuda.empty_cache()
    try:
        out.update(config5_leg(dev, torch, ops, leg))
    except Exception as e:   # (kept out of the headline's way)
        out["config5"] = {"error": repr(e)}
    return out


def config5_leg(dev, torch, ops, leg):
    """config 5: MedSAM ViT-B + a 4-class prototype bank on 1024x1024 slices: the query is encoded ONCE by DINOv2 at 1022^2 and matched
    against the four banks (validation.py:207: four 1-way passes sharing one encoder forward), then MedSAM per class."""
    from protosam_amd.synth_cases import cfg5_inputs   # (input generator only: seeded synthetic 4-organ slice)
    from protosam_amd.grid_proto_fewshot import FewShotSeg
    from protosam_amd.protomedsam import ProtoMedSAM
    from protosam_amd.protosam import ALPNetWrapper
    from protosam_amd.runner import ALP_CFG
    from protosam_amd.synth import synth_state_dict
    S = 1024
    alp = FewShotSeg(S, None, dict(ALP_CFG))
    alp.load_state_dict(synth_state_dict(alp, 1234))
    alp = alp.to(dev).eval()
    model = ProtoMedSAM((1024, 1024), ALPNetWrapper(alp), "random:vit_b:1234", use_cca=True).to(dev).eval()
    s_img, s_masks, q_img = cfg5_inputs()
    s_img, q_img = s_img.to(dev), q_img.to(dev)
    s_masks = [m.to(dev) for m in s_masks]
    res = {}
    holder = {}

    def run():
        holder["out"] = model.forward_classes(q_img, s_img, s_masks)
    leg("config5", run, 1, 3, unit="slices/s", library_default=True,
        note="one 1024x1024 slice, 4 classes: one DINOv2 forward of the query at 1022^2 shared by the four prototype banks, MedSAM "
             "ViT-B image encoder once, box-prompted decoder per class (ProtoMedSAM.forward_classes)")
    return res


def cpu_baseline(model, alp_sd, vol, svol, slab, args, dev):
    """CPU oracle (kind 'port': our restatement, pinned against the reference) on a bounded sample of the same workload:
    `--cpu-slices` slices with all host threads (capped at 32) and `--cpu-1thread-slices` at one thread, which is how the
    reference runs (`torch.set_num_threads(1)`, validation_protosam.py:299). Doubles as a full-depth parity check."""
    import numpy as np
    import torch
    from oracle import alp as oalp, dinov2 as odino, glue
    from protosam_amd.metrics import dice
    from protosam_amd.runner import part_assign, run_slices, support_set
    # all host cores up to 32: beyond that PyTorch's intra-op parallelism stops scaling on these layer sizes
    cores = min(os.cpu_count() or 1, 32)
    n = args.slices
    zs_all = [int((i + 0.5) * n / max(args.cpu_slices, 1)) for i in range(args.cpu_slices)]
    sup_imgs, sup_masks = support_set(svol, slab)
    enc_sd = {k[len("encoder."):]: v for k, v in alp_sd.items() if k.startswith("encoder.")}
    sam_sd = {k: v.detach().cpu().float() for k, v in model.sam.state_dict().items()}
    enc = lambda im: odino.forward_features(im, enc_sd, "dinov2_b14")["x_norm_patchtokens"]  # noqa: E731

    def oracle_slice(z, taps):
        q = vol[z][None, None].repeat(1, 3, 1, 1).contiguous()
        part = part_assign(z, n)
        with torch.no_grad():
            logits = oalp.fewshot_forward(enc, sup_imgs[part], sup_masks[part], q, 512)
            return glue.protosam_forward(q, logits, sam_sd, args.sam, use_bbox=True, use_points=True, point_mode="both",
                                         use_cca=False, encoder_depth=args.cpu_sam_depth, taps=taps)
    sup_d = [s.to(dev) for s in sup_imgs]
    msk_d = [m.to(dev) for m in sup_masks]
    vol_d = vol.to(dev)
    torch.set_num_threads(cores)

    def gpu_parity(z, pred_ref, scores_ref, taps):
        # GPU result for the same slice THROUGH THE TIMED PATH: a micro-batch of the slice's z-part, as the steps above run them
        part_z = [y for y in range(args.slices) if part_assign(y, args.slices) == part_assign(z, args.slices)]
        i = part_z.index(z)
        zs_b = part_z[max(0, min(i - args.micro // 2, len(part_z) - args.micro)):][:args.micro]
        masks, _ = run_slices(model, vol_d, sup_d, msk_d, zs_b, dev, batch=args.micro)
        b = zs_b.index(z)
        g, r = masks[b].cpu().float(), pred_ref.float()
        st = model.last_stats
        span = next((sp for sp in st.get("spans", []) if sp[0] == b), None) if args.micro > 1 else (0, 0, st.get("n_prompts", 0))
        p = {"slice": z, "dice_final_mask": round(dice(g, r), 5), "flipped_pixels": int((g != r).sum()),
             "components": int(span[2]) if span else 0, "slices_in_call": len(zs_b)}
        if "low_res" in st and span and len(taps.get("low_res", [])) == span[2]:
            low = st["low_res"][span[1]:span[1] + span[2], st["sel"]].cpu()
            low_ref = torch.stack([l[0] for l in taps["low_res"]])
            p["max_abs_dprob_low_res"] = float((torch.sigmoid(low) - torch.sigmoid(low_ref)).abs().max())
            p["max_abs_dscore"] = float(np.abs(st["iou"][span[1]:span[1] + span[2], st["sel"]].cpu().numpy() - np.array(scores_ref)).max())
        return p
    t_all, parities, parities_x3 = 0.0, [], []
    sam_enc = model.sam.image_encoder
    for z in zs_all:
        taps = {}
        t0 = time.perf_counter()
        pred_ref, scores_ref = oracle_slice(z, taps)
        t_all += time.perf_counter() - t0
        parities.append(gpu_parity(z, pred_ref, scores_ref, taps))
        if not args.no_extras:        # the same slice with the SAM encoder's Linear layers at the reference's width (`reference_width`)
            sam_enc.gemm_x3 = True
            try:
                parities_x3.append(gpu_parity(z, pred_ref, scores_ref, taps))
            finally:
                sam_enc.gemm_x3 = False
    log(f"cpu_baseline: {len(zs_all)} slices in {t_all:.1f}s on {cores} threads")
    cpu = {"value": round(len(zs_all) / t_all, 5), "unit": "slices/s", "cores": cores, "kind": "port",
           "sample": f"{len(zs_all)} slices (z = {zs_all}) of the same volume through the full CPU oracle pipeline "
                     f"(fp32, {cores} threads), {t_all:.1f} s"}
    if args.cpu_1thread_slices > 0:
        torch.set_num_threads(1)
        t0 = time.perf_counter()
        for z in zs_all[:args.cpu_1thread_slices]:
            oracle_slice(z, {})
        t1 = time.perf_counter() - t0
        torch.set_num_threads(cores)
        log(f"cpu_baseline: {args.cpu_1thread_slices} slice(s) in {t1:.1f}s on 1 thread")
        cpu["one_thread"] = {"value": round(args.cpu_1thread_slices / t1, 5), "unit": "slices/s", "cores": 1,
                             "sample": f"{args.cpu_1thread_slices} slice(s), torch.set_num_threads(1) as "
                                       f"validation_protosam.py:299, {t1:.1f} s"}
    worst = max((p.get("max_abs_dprob_low_res", 0.0) for p in parities), default=0.0)
    parity = {"slices": parities, "worst_max_abs_dprob_low_res": worst, "bound": 1e-3,
              "min_dice_final_mask": min(p["dice_final_mask"] for p in parities)}
    if parities_x3:
        parity["reference_width"] = {"slices": parities_x3, "bound": 1e-3,
                                     "worst_max_abs_dprob_low_res": max((p.get("max_abs_dprob_low_res", 0.0) for p in parities_x3), default=0.0),
                                     "min_dice_final_mask": min(p["dice_final_mask"] for p in parities_x3)}
    return cpu, parity


if __name__ == "__main__":
    main()
