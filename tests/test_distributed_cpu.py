"""The N>1 path on CPU: world_size-2 gloo processes run the same code bench.py runs per rank
(weight-bundle broadcast from rank 0, contiguous image sharding, max-over-ranks timing
reduction) — everything except the HIP calls."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lrp_imagecaptioning_amd.parallel import broadcast_weights, shard_range


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
    shapes = {"block1_conv1_W": (3, 3, 3, 8), "block1_conv1_b": (8,), "lstm_Wi": (16, 32), "V": (8, 1)}
    w0 = None
    if rank == 0:
        rs = np.random.RandomState(42)
        w0 = {k: rs.standard_normal(s).astype(np.float32) for k, s in shapes.items()}
    got = broadcast_weights(w0, shapes, 0, dist)
    checksum = float(sum(float(v.double().sum()) for v in got.values()))
    lo, hi = shard_range(10, world, rank)
    t = torch.tensor([0.5 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)          # bench.py: max over ranks of the timed region
    q.put((rank, checksum, lo, hi, float(t.item()), {k: tuple(v.shape) for k, v in got.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_and_sharding_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, c0, lo0, hi0, t0, s0), (r1, c1, lo1, hi1, t1, s1) = res
    assert c0 == c1 and s0 == s1                       # every rank holds the same bundle
    assert s0["block1_conv1_W"] == (3, 3, 3, 8)
    assert (lo0, hi0, lo1, hi1) == (0, 5, 5, 10)       # contiguous, disjoint, complete
    assert t0 == t1 == 1.5                             # MAX over ranks
