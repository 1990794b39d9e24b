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
