"""GPU, world_size 2 on ONE device: the multi-rank path with LIVE models - each rank builds its own ProtoSAM, runs its shard of a volume
through `run_slices` (real kernels), the uint8 masks are all-gathered and re-ordered, and the result equals what one process computes for
the same shards. RCCL refuses two ranks on one device, so the collective here is gloo with the device tensors staged through the host
(`runner.all_gather_rows`); everything else - sharding, per-rank models and workspaces, the gather's row order, `bench.py`'s rank body with
real steps - is the code an 8-GPU job runs. (tests/test_runner_dist_cpu.py covers the same logic with stubbed masks.)"""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_slices, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from protosam_amd.runner import build_protosam, gather_masks, interleave_rank_major, run_slices, shard_slices, support_set
    from protosam_amd.synth import synth_volume
    dev = torch.device("cuda:0")
    model, _ = build_protosam(dev, sam_type="vit_b", image_size=512, dino_depth=2, sam_depth=2)
    vol, _ = synth_volume(n_slices, 512, seed=0, kind="ct")
    svol, slab = synth_volume(n_slices, 512, seed=1, kind="ct")
    vol_d = vol.to(dev)
    sup_imgs, sup_masks = support_set(svol.to(dev), slab.to(dev))
    zs = shard_slices(n_slices, rank, world)
    masks, st = run_slices(model, vol_d, sup_imgs, sup_masks, zs, dev, batch=len(zs))
    full = gather_masks(masks, world)
    got = interleave_rank_major(full, n_slices, world).cpu()
    res = dict(rank=rank, prompts=st, fg=[int(m.sum()) for m in masks.cpu()])
    if rank == 0:       # the same shards, one after the other, in this process: identical launches, so identical masks
        ref = torch.zeros((n_slices, 512, 512), dtype=torch.uint8)
        for r in range(world):
            zr = shard_slices(n_slices, r, world)
            m, _ = run_slices(model, vol_d, sup_imgs, sup_masks, zr, dev, batch=len(zr))
            ref[torch.tensor(zr)] = m.cpu()
        res["equal"] = bool(torch.equal(got, ref))
        res["nonempty"] = int((ref.flatten(1).sum(1) > 0).sum())
    q.put(res)
    dist.destroy_process_group()


def test_two_ranks_live_models_gather_equals_one_process(dev):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 8, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    r0 = next(r for r in res if r["rank"] == 0)
    print({k: v for k, v in r0.items() if k != "rank"}, [r["prompts"] for r in res])
    assert r0["equal"] and r0["nonempty"] >= 6
    assert all(len(r["prompts"]) == 4 for r in res)


def test_bench_rank_body_world2_on_one_gpu(dev):
    """`bench.py` itself under torch.distributed.run with two ranks sharing the GPU (PSAM_BENCH_BACKEND=gloo, PSAM_BENCH_SHARE_GPU=1):
    the N > 1 JSON line of a real run - `ranks_seen`, per-rank times, the all-gather's time - with truncated models so that it takes
    seconds. A functional check; its throughput means nothing."""
    env = dict(os.environ, PSAM_BENCH_BACKEND="gloo", PSAM_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4",
           "--micro", "4", "--slices", "24", "--sam", "vit_b", "--cpu-sam-depth", "2", "--no-cpu-baseline"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{") and '"metric"' in l][-1])
    print({k: line[k] for k in ("value", "n_gpus", "ranks_seen", "per_rank_ms_per_step", "allgather_ms_per_step", "collective_backend")})
    assert line["n_gpus"] == 2 and line["ranks_seen"] == [0, 1] and line["collective_backend"] == "gloo"
    assert len(line["per_rank_ms_per_step"]) == 2 and all(v > 0 for v in line["per_rank_ms_per_step"])
    assert len(line["allgather_ms_per_step"]) == 2 and all(v > 0 for v in line["allgather_ms_per_step"])
    assert abs(line["value"] * line["ms_per_step"] * 1e-3 - 8) < 0.1 and line["cpu_baseline"] is None
    assert line["config"]["mean_components_per_slice"] > 0
