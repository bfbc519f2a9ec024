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


def _train_worker(rank, world, port, q):
    """Data-parallel fine-tune step: each rank differentiates its half of the batch (oracle arithmetic stands in for the
    HIP step), `average_gradients` makes them the full-batch gradient."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lrp_imagecaptioning_amd.parallel import average_gradients
    from lrp_imagecaptioning_amd.synthetic import adaptive_weights, vgg_weights
    from oracle import train_ref as T
    cfg = [("c1", 3, 8, True), ("c2", 8, 8, False)]
    HW, L, D, H, V, B, Tn = 8, 16, 8, 8, 12, 4, 4
    rs = np.random.RandomState(5)
    w = vgg_weights(rs, cfg, bias_std=0.3)
    w.update(adaptive_weights(rs, L, D, H, H, V))
    X = rs.uniform(-2, 2, size=(B, HW, HW, 3)).astype(np.float32)
    cap_in = rs.randint(0, V, size=(B, Tn))
    y = rs.randint(0, V, size=(B, Tn))
    lw = 1 + rs.uniform(0, 1, size=(B, Tn, V))
    names = T.param_names(cfg)
    flat = lambda g: torch.from_numpy(np.concatenate([g[k].ravel() for k in names]))
    lo, hi = shard_range(B, world, rank)
    tot, l1, l2, g, _ = T.loss_and_grads(w, cfg, X[lo:hi], cap_in[lo:hi], y[lo:hi], lw[lo:hi])
    mine, losses = average_gradients(flat(g), torch.tensor([tot, l1, l2], dtype=torch.float64))
    full = T.loss_and_grads(w, cfg, X, cap_in, y, lw)
    err = float((mine - flat(full[3])).abs().sum() / flat(full[3]).abs().sum())
    q.put((rank, err, float(losses[0]), full[0], float(mine.double().sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_average_world2_equals_full_batch():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_train_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err, loss, full_loss, _ in res:
        assert err < 1e-12 and abs(loss - full_loss) < 1e-12
    assert res[0][4] == res[1][4]                      # both ranks step with the same gradient


# ---------------------------------------------------------------------------------------------- world 4 and 8, ragged shards
def _big_worker(rank, world, port, q, n_images, vocab):
    """bench.py's start-up at its REAL size: the flat frozen bundle of VGG16 + adaptive attention at V = 10 000
    (23.7 M floats = 95 MB, ONE broadcast from rank 0), then the contiguous image shards of a batch that does not divide
    by the world size (config 3's 256 images shard evenly; 250 do not), the all-gather of every rank's own clock and the
    MAX-over-ranks reduction of the timed region."""
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shapes = bench.synth_weights_shapes(vocab)
    w0 = None
    if rank == 0:                                       # cheap deterministic fill: the test is about the transport
        w0 = {k: (np.arange(int(np.prod(s)), dtype=np.float32) % 251 + i).reshape(s) for i, (k, s) in enumerate(sorted(shapes.items()))}
    got = broadcast_weights(w0, shapes, 0, dist)
    nbytes = 4 * sum(int(v.numel()) for v in got.values())
    ok = all(tuple(got[k].shape) == tuple(s) for k, s in shapes.items())
    chk = 0.0
    for i, (k, s) in enumerate(sorted(shapes.items())):  # every rank recomputes what rank 0 filled in
        n = int(np.prod(s))
        want = float(((np.arange(n, dtype=np.float64) % 251) + i).sum())
        chk += abs(float(got[k].double().sum()) - want)
    lo, hi = shard_range(n_images, world, rank)
    t = torch.tensor([1.0 + 0.25 * rank], dtype=torch.float64)
    every = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(every, t)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    q.put((rank, nbytes, ok, chk, lo, hi, float(t.item()), [float(e.item()) for e in every]))
    dist.barrier()
    dist.destroy_process_group()


def _run_big(world, n_images=250, vocab=10000):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_big_worker, args=(r, world, port, q, n_images, vocab)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


def _check_big(res, world, n_images):
    assert [r[0] for r in res] == list(range(world))
    for rank, nbytes, ok, chk, lo, hi, tmax, every in res:
        assert nbytes >= 94e6 and ok and chk == 0.0, (rank, nbytes, ok, chk)       # the 95 MB bundle, bit for bit
        assert tmax == 1.0 + 0.25 * (world - 1)                                    # MAX over ranks
        assert every == [1.0 + 0.25 * r for r in range(world)]                     # every rank's own clock, in rank order
    # shards: contiguous, disjoint, complete, sizes differ by at most one, the larger ones first
    spans = [(r[4], r[5]) for r in res]
    assert spans[0][0] == 0 and spans[-1][1] == n_images
    assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
    sizes = [hi - lo for lo, hi in spans]
    assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True) and sum(sizes) == n_images


def test_real_size_bundle_and_ragged_shards_world4():
    _check_big(_run_big(4), 4, 250)


def test_real_size_bundle_and_ragged_shards_world8():
    """BASELINE configs[2]'s rank count (8 GPUs of one node) on the CPU: 250 images -> 32, 32, 31, 31, 31, 31, 31, 31."""
    res = _run_big(8)
    _check_big(res, 8, 250)
    assert [r[5] - r[4] for r in res] == [32, 32, 31, 31, 31, 31, 31, 31]


def test_shard_range_properties():
    for world in (1, 2, 3, 4, 8):
        for n in (0, 1, 7, 8, 31, 32, 250, 256):
            spans = [shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1 and all(s >= 0 for s in sizes)
