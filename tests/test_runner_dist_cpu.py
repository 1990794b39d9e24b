"""CPU / gloo, world_size 2: slice sharding z = r (mod W) and the single all-gather of uint8 masks reassemble the
volume exactly as a 1-rank run would produce it (the N > 1 path of bench.py / protosam_amd.runner)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_mask(z, S=16):
    g = torch.Generator().manual_seed(1000 + z)
    return (torch.rand((S, S), generator=g) > 0.5).to(torch.uint8)


def _worker(rank, world, port, n_slices, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from protosam_amd.runner import gather_masks, interleave_rank_major, shard_slices
    zs = shard_slices(n_slices, rank, world)
    k = -(-n_slices // world)
    local = torch.zeros((k, 16, 16), dtype=torch.uint8)
    for i, z in enumerate(zs):
        local[i] = _fake_mask(z)
    full = gather_masks(local, world)
    vol = interleave_rank_major(full, n_slices, world)
    ok = all(torch.equal(vol[z], _fake_mask(z)) for z in range(n_slices))
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)  # the max-over-ranks timing reduction used by bench.py
    q.put((rank, ok, float(t.item()), zs))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_slices", [8, 7])
def test_two_rank_gather_matches_single_rank(n_slices):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_slices, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, tmax, zs in res:
        assert ok and tmax == 2.0
    shards = sorted(z for r in res for z in r[3])
    assert shards == list(range(n_slices))  # disjoint cover


def test_part_assignment_and_support_set():
    from protosam_amd.runner import part_assign, support_set
    assert [part_assign(z, 9) for z in range(9)] == [0, 0, 0, 1, 1, 1, 2, 2, 2]
    vol = torch.arange(9.0).view(9, 1, 1).expand(9, 4, 4).contiguous()
    imgs, masks = support_set(vol, vol)
    assert [int(i[0, 0, 0, 0]) for i in imgs] == [1, 4, 7] and imgs[0].shape == (1, 3, 4, 4) and masks[0].shape == (1, 4, 4)


def _bench_worker(rank, world, port, B, n_slices, steps, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from protosam_amd.runner import gather_masks, part_assign
    parts = [[z for z in range(n_slices) if part_assign(z, n_slices) == pt] for pt in range(3)]
    ok = True
    for s in range(steps):
        zs = bench.step_slices(s, parts, B, world, rank)
        local = torch.stack([_fake_mask(z) for z in zs])
        full = gather_masks(local, world)                                  # rank-major [world*B, S, S]
        all_zs = [z for r in range(world) for z in bench.step_slices(s, parts, B, world, r)]
        one_rank = bench.step_slices(s, parts, B * world, 1, 0)            # the same window on ONE rank with batch world*B
        ok &= sorted(all_zs) == sorted(one_rank)
        ok &= all(torch.equal(full[i], _fake_mask(z)) for i, z in enumerate(all_zs))
        ok &= len({z for z in zs}) == len(zs) or len(parts[s % 3]) < B * world   # no slice twice unless the part wraps
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_bench_step_indexing_world2_equals_single_rank():
    """bench.py's N > 1 step indexing: the window of step s gathered from 2 ranks (z = r mod 2) holds exactly the slices a
    1-rank run with twice the batch processes in step s, each rank's rows where the all-gather puts them."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, 8, 64, 7, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res)


def test_bench_launches_its_own_ranks(monkeypatch, capsys):
    """`python bench.py --gpus N` without a launcher starts N ranks through torch.distributed.run BEFORE touching the GPU and
    relays rank 0's line (the command is intercepted here: there is no GPU in the build container)."""
    import subprocess
    import sys
    import bench
    seen = {}

    def fake_run(cmd, env=None, stdout=None, text=None):
        seen["cmd"], seen["env"] = cmd, env
        return subprocess.CompletedProcess(cmd, 0, stdout='noise\n{"metric": "m", "n_gpus": 4}\n')
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "127.0.0.1" in cmd
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert capsys.readouterr().out.strip() == '{"metric": "m", "n_gpus": 4}'
    assert "torch.cuda" not in str(getattr(bench, "__dict__", {}).get("torch", ""))   # parent path imports no torch at module level


def _strong_worker(rank, world, port, n_slices, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from protosam_amd.runner import gather_masks, interleave_rank_major
    zs = bench.strong_slices(n_slices, world, rank)                       # one pass over the whole volume per step
    local = torch.stack([_fake_mask(z) for z in zs])
    full = gather_masks(local, world)
    vol = interleave_rank_major(full, n_slices, world)
    ok = all(torch.equal(vol[z], _fake_mask(z)) for z in range(n_slices)) and len(zs) == n_slices // world
    q.put((rank, bool(ok), zs))
    dist.destroy_process_group()


def test_bench_strong_scaling_indexing_world2():
    """`bench.py --scaling strong`: the 64-slice volume split z = r (mod W), one all-gather per step reassembles it in z order."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_strong_worker, args=(r, 2, port, 64, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert sorted(z for r in res for z in r[2]) == list(range(64))
    import bench
    assert bench.strong_slices(64, 8, 3) == [3, 11, 19, 27, 35, 43, 51, 59]


class _HostEvent:
    def record(self):
        import time
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


class _NullTimer:
    def summary(self):
        return 0, 0.0, 0.0

    def by_tag(self):
        return {}


def _fake_mask512(z):
    g = torch.Generator().manual_seed(7000 + z)
    return (torch.rand((512, 512), generator=g) > 0.5).to(torch.uint8)


class HostRuntime:
    """bench.DeviceRuntime's interface on the host: gloo instead of RCCL, wall-clock events, no model - `run_slices` hands back a
    deterministic mask per slice index, so what the all-gather moved can be checked row by row."""
    backend = "gloo"

    def device(self, local_rank):
        return torch.device("cpu")

    def init_group(self, local_rank):
        dist.init_process_group("gloo")

    def sync(self):
        pass

    def event(self):
        return _HostEvent()

    def kernel_timer(self):
        return _NullTimer()

    def power_sampler(self, local_rank):
        return None

    def build(self, args, dev):
        return None, {}, None, None, None, torch.zeros((args.slices, 1, 1)), [None] * 3, [None] * 3

    def run_slices(self, model, vol, sup_imgs, sup_masks, zs, dev, out=None, batch=1):
        if out is None:
            out = torch.zeros((len(zs), 512, 512), dtype=torch.uint8)
        for i, z in enumerate(zs):
            out[i] = _fake_mask512(z)
        return out[:len(zs)], [1 + (z % 2) for z in zs]


def _bench_main_worker(rank, world, port, argv, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import bench
    r = bench.main(argv, runtime=HostRuntime())
    B = r["last_gather"].shape[0] // world
    rows_ok = all(torch.equal(r["last_gather"][rank * B + i], _fake_mask512(z)) for i, z in enumerate(r["last_zs"]))
    q.put((rank, r["last_zs"], rows_ok, r.get("line"), int(r["last_gather"].shape[0]), torch.get_num_threads(), dist.is_initialized()))


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_main_rank_body_world2(scaling):
    """bench.main()'s rank body - everything between the launcher and the JSON line - at world 2 under gloo with a stubbed `run_slices`
    (bench.DeviceRuntime -> HostRuntime): argument handling, the all-gather of rank ids, warm-up, the timed loop with the event pairs
    around the all-gather, the per-rank reductions, rank 0's line and the teardown. What the first real 8-GPU run can then still find
    is RCCL and the kernels, not bench.py."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    argv = ["--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "4", "--micro", "4", "--slices", "24", "--scaling", scaling]
    procs = [ctx.Process(target=_bench_main_worker, args=(r, 2, port, argv, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    B = 12 if scaling == "strong" else 4
    (r0, zs0, ok0, line, rows0, thr0, init0), (r1, zs1, ok1, line1, rows1, thr1, init1) = res
    assert ok0 and ok1 and rows0 == rows1 == 2 * B and line1 is None and not init0 and not init1       # (process groups destroyed)
    assert not set(zs0) & set(zs1) and len(zs0) == len(zs1) == B
    if scaling == "strong":
        assert sorted(zs0 + zs1) == list(range(24))
    assert thr0 == thr1 == max(1, (os.cpu_count() or 1) // 2)
    assert line["n_gpus"] == 2 and line["ranks_seen"] == [0, 1] and line["scaling"] == scaling and line["steps"] == 3
    assert len(line["per_rank_ms_per_step"]) == 2 and len(line["allgather_ms_per_step"]) == 2
    assert abs(max(line["per_rank_ms_per_step"]) - line["ms_per_step"]) < 1e-2
    assert abs(line["value"] * line["ms_per_step"] * 1e-3 - 2 * B) < 1e-2 * 2 * B          # whole-job slices per step / max-over-ranks time
    assert line["cpu_baseline"] is None and "per_slice_forward" not in line and line["config"]["slices_per_step_per_gpu"] == B
