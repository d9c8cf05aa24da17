"""N>1 path on CPU: world_size-2 gloo processes exercise the weight broadcast and the frame
sharding used by bench.py (the data path itself has no collective)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import bench_support as bs


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(0)
    blobs = {0: rng.integers(0, 256, 100003, dtype=np.uint8).tobytes(),
             1: rng.integers(0, 256, 4097, dtype=np.uint8).tobytes()} if rank == 0 else None
    got, ms = bs.broadcast_blobs(blobs, torch.device("cpu"), rank)
    mine = bs.shard_frames(13, rank, world)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    # max-over-ranks timing reduction exactly as bench.py does it
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    q.put((rank, {k: (len(v), sum(v[:64])) for k, v in got.items()}, gathered, float(t.item())))
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_and_sharding_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    res.sort()
    assert res[0][1] == res[1][1] and res[0][1][0][0] == 100003 and res[0][1][1][0] == 4097
    shards = res[0][2]
    assert sorted(shards[0] + shards[1]) == list(range(13)) and not set(shards[0]) & set(shards[1])
    assert res[0][3] == res[1][3] == 2.0


def _run_bench(args, env_extra=None, timeout=240):
    import json
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, capture_output=True, text=True, env=env,
                       timeout=timeout, cwd=root)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None), lines


def test_bench_self_spawns_n_ranks_dry_run():
    """`python bench.py --gpus 2` with no launcher must itself start 2 ranks (torch.distributed.run) and report
    n_gpus = 2 on ONE stdout JSON line; the CPU tier drives exactly that launch path with --dry-run-cpu (gloo)."""
    r, out, lines = _run_bench(["--gpus", "2", "--dry-run-cpu", "--frames", "7", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, r.stdout
    assert out["n_gpus"] == 2 and out["dry_run"] is True and out["value"] is None
    assert out["frames_per_rank"] == [7, 7] and out["max_over_ranks_check"] == 2.0


def test_bench_world8_dry_run_launched_like_the_driver():
    """The node shape the driver measures: `python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1
    --master-port P bench.py --gpus 8 ...`, eight ranks, 96 frames each (frame f of the job -> rank f mod 8), one JSON line from rank 0."""
    import json
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["OMP_NUM_THREADS"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "8", "--dry-run-cpu", "--frames", "96",
           "--steps", "2", "--warmup", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=420, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["dry_run"] is True and out["value"] is None and out["scaling"] == "weak"
    assert out["frames_per_rank"] == [96] * 8 and out["max_over_ranks_check"] == 8.0
    shards = [bs.shard_frames(96 * 8, rk, 8) for rk in range(8)]
    assert sorted(f for sh in shards for f in sh) == list(range(768)) and all(len(sh) == 96 for sh in shards)
    assert all(f % 8 == rk for rk, sh in enumerate(shards) for f in sh)


def test_bench_refuses_more_gpus_than_visible():
    """--gpus N on a node with fewer GPUs fails loudly (non-zero exit, no JSON line) instead of printing a 1-GPU line."""
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("2+ GPUs visible")
    r, out, lines = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0 and not lines
    assert "--gpus 2" in (r.stderr + r.stdout)


def test_bench_refuses_world_size_mismatch():
    r, out, lines = _run_bench(["--gpus", "2", "--dry-run-cpu"], env_extra={"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and not lines
