"""CPU, world_size 2 over gloo: the N>1 path of the hot path = batch sharding + the single all-gather
of the final keypoints (what bench.py --gpus N runs over RCCL)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, global_batch, q):
    from sgtapose_amd.distributed import gather_keypoints, shard_bounds, shard_inputs
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(3)
        full = {"kp": torch.randn(global_batch, 7, 3, generator=g), "x": torch.randn(global_batch, 3, 8, 8, generator=g)}
        mine = shard_inputs(full, rank, world)
        lo, hi = shard_bounds(global_batch, rank, world)
        assert mine["x"].shape[0] == hi - lo and torch.equal(mine["kp"], full["kp"][lo:hi])
        # stand-in for the per-rank HIP path: a deterministic function of the local shard
        kp_local = mine["kp"] * 2 + 1
        out = gather_keypoints(kp_local, global_batch)
        ok = torch.equal(out, full["kp"] * 2 + 1)
        q.put((rank, bool(ok), tuple(out.shape)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("global_batch", [8, 7])
def test_shard_and_allgather_world2(global_batch):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, global_batch, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] for r in res) and all(r[2] == (global_batch, 7, 3) for r in res)


def test_shard_bounds_cover_batch_exactly():
    from sgtapose_amd.distributed import shard_bounds
    for gb in (1, 7, 32, 256):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(gb, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == gb
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(8, 2, 2)


@pytest.mark.parametrize("n,extra", [(2, []), (2, ["--dtype", "bf16"]), (2, ["--dtype", "f16"]), (2, ["--dtype", "f32x3"]),
                                     (4, ["--arch", "dlapawdT_34", "--frames", "4"])])
def test_bench_self_launches_ranks_and_gathers(n, extra):
    """``python bench.py --gpus N`` (no torchrun) for the configurations that run multi-GPU (configs[2]: bf16 shards;
    configs[4]: T = 4 windows over 4 ranks): the parent -- which makes no GPU call -- spawns the ranks, rank 0 prints ONE
    JSON line, every rank holds the same gathered key-points, exit code 0.  --dry-run keeps it on the CPU (gloo) -- the
    launcher, rendezvous and all-gather flow are those of the GPU run."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", str(n), "--dry-run", "--batch", "3",
                        "--steps", "2", "--warmup", "1"] + extra, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["ranks_seen"] == n and rec["n_gpus"] == n and rec["gather_ok"] is True
    assert rec["gathered_keypoints_equal_on_all_ranks"] is True


def test_bench_under_the_torch_launcher():
    """The driver's form: python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2 (ranks from the launcher)."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(_free_port()), os.path.join(repo, "bench.py"), "--gpus", "2",
                        "--dry-run", "--batch", "2", "--steps", "2", "--warmup", "1"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["ranks_seen"] == 2
