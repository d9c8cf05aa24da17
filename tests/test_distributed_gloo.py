"""N>1 path on CPU: world_size-2 gloo processes exercise the weight broadcast and the frame
sharding used by bench.py (the data path itself has no collective)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from peppa_pig_face_landmark_amd import bench_support as bs


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
