"""Multi-GPU support for the hot path.  The path shards by image:
every (image, token) unit is independent once the frozen weights are resident, so
the only collective is ONE broadcast of the weight bundle at start-up (RCCL over
xGMI when the backend is "nccl"; gloo in the CPU tests).  No data-path exchange."""
import numpy as np
import torch


def shard_range(n_items, world, rank):
    """Contiguous shard [lo, hi) of n_items for `rank` (an image's tokens stay on one GPU,
    they share its cached gates).  Ragged tails go to the lowest ranks."""
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def flatten_bundle(weights, shapes):
    """dict -> one flat float32 vector in the canonical (sorted-name) order."""
    parts = []
    for name in sorted(shapes):
        a = np.ascontiguousarray(weights[name], dtype=np.float32)
        assert tuple(a.shape) == tuple(shapes[name]) or a.size == int(np.prod(shapes[name])), name
        parts.append(a.reshape(-1))
    return np.concatenate(parts)


def unflatten_bundle(flat, shapes):
    out, off = {}, 0
    for name in sorted(shapes):
        n = int(np.prod(shapes[name]))
        out[name] = flat[off:off + n].reshape(shapes[name])
        off += n
    assert off == flat.numel() if isinstance(flat, torch.Tensor) else off == flat.size
    return out


def broadcast_weights(weights_rank0, shapes, device_index, dist, src=0):
    """One bucket, one collective: rank `src` flattens the bundle, everyone receives it.
    Returns dict name -> tensor view (on the GPU for nccl, on the CPU for gloo)."""
    total = sum(int(np.prod(s)) for s in shapes.values())
    on_gpu = dist.get_backend() == "nccl"
    dev = torch.device("cuda", device_index) if on_gpu else torch.device("cpu")
    if dist.get_rank() == src:
        flat = torch.from_numpy(flatten_bundle(weights_rank0, shapes)).to(dev)
    else:
        flat = torch.empty(total, dtype=torch.float32, device=dev)
    dist.broadcast(flat, src=src)
    return unflatten_bundle(flat, shapes)


def average_gradients(flat_grads, losses=None, group=None):
    """Data-parallel fine-tune step: every rank holds the flat gradient of ITS shard's mean loss; with
    equal shards the global-batch gradient is the mean over ranks.  One all-reduce over the whole flat buffer (a single
    ~125 MB bucket for VGG16 + decoder: on xGMI rings few large transfers beat many small ones), in place."""
    dist = torch.distributed
    if not (dist.is_available() and dist.is_initialized()):
        return flat_grads, losses
    world = dist.get_world_size(group)
    if world == 1:
        return flat_grads, losses
    dist.all_reduce(flat_grads, group=group)
    flat_grads /= world
    if losses is not None:
        dist.all_reduce(losses, group=group)
        losses /= world
    return flat_grads, losses
