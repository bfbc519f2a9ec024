#!/usr/bin/env python3
"""BASELINE config 5 on ONE GPU's share: the LRP-inference fine-tune step (train.py:571-580) of the VGG16 +
adaptive-attention captioner — predict, lrp_weight for every word of every predicted caption, gradients, Adam.
The config quotes batch=64 on 8 GPUs = 8 images per GPU; B=32 (the reference's config.batch_size) is timed as well.
Prints the phases; not the headline bench."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from lrp_imagecaptioning_amd.explainers import (CaptionModelSpec, ExplainImgCaptioningAdaptiveAttention,
                                                    ExplainImgCaptioningGridTDModel)
    from lrp_imagecaptioning_amd.synthetic import adaptive_weights, gridtd_weights, images, vgg_weights
    from lrp_imagecaptioning_amd.training import TrainingLRPInferenceAdaptive, TrainingLRPInferenceGridTD
    B, T, V = int(os.environ.get("B", 8)), int(os.environ.get("T", 21)), 10000
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:                                       # data-parallel rehearsal: B images per rank, one gradient all-reduce
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local)
        backend = os.environ.get("BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    kind = os.environ.get("DEC", "adaptive")            # DEC=gridtd: the grid-TD twin (train.py:596-669)
    rs = np.random.RandomState(0)                       # same weights on every rank (a real run broadcasts them)
    w = vgg_weights(rs)
    w.update((adaptive_weights if kind == "adaptive" else gridtd_weights)(rs, 196, 512, 512, 512, V))
    spec = CaptionModelSpec(w, img_encoder="vgg16", hidden_dim=512, embedding_dim=512, L=196, D=512, vocab_size=V)
    if kind == "adaptive":
        ex = ExplainImgCaptioningAdaptiveAttention(spec, None, None, max_caption_length=T - 1, max_images=B)
        tr = TrainingLRPInferenceAdaptive(ex, learning_rate=2e-4, drop_rate=0.5)
    else:
        ex = ExplainImgCaptioningGridTDModel(spec, None, None, max_caption_length=T - 1, max_images=B)
        tr = TrainingLRPInferenceGridTD(ex, learning_rate=2e-4, drop_rate=0.5)
    eng = ex._engine
    eng.train_set_precision(os.environ.get("TRAIN_PREC", "bf16"))   # BASELINE config 5 names bf16; TRAIN_PREC=fp32: the fp32-grade step
    rs = np.random.RandomState(100 + rank)              # this rank's shard of the batch
    X = torch.as_tensor(images(rs, B)).cuda()
    cap_in = np.concatenate([np.full((B, 1), 1), rs.randint(2, V, size=(B, T - 1))], axis=1).astype(np.int32)
    y = rs.randint(0, V, size=(B, T)).astype(np.int32)

    def timed(fn):
        torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e3, r
    for _ in range(3):                                                  # warm-up (lazy allocations; the decoder scans are captured as hipGraphs on their second run)
        tr.train_on_batch([cap_in, X], y)
    t_pred, y_pred = timed(lambda: tr.predict_on_batch([cap_in, X]))
    t_lrp, lw_dev = timed(lambda: tr._lrp_layer.call_device(X, y_pred, images_encoded=True))
    n_maps = int((lw_dev != 1).sum())
    masks = tr._masks(B, T)
    t_step, (g, losses) = timed(lambda: eng.train_step(cap_in, y, lw_dev, masks, grads=tr._grads))
    from lrp_imagecaptioning_amd.parallel import average_gradients
    t_ar, _ = timed(lambda: average_gradients(g, losses))          # (no-op at world size 1)
    t_apply, _ = timed(lambda: eng.train_apply(g))
    n = int(os.environ.get("N", 6))
    t_all, _ = timed(lambda: [tr.train_on_batch([cap_in, X], y) for _ in range(n)])
    if world > 1:
        import torch.distributed as dist
        flat = tr._engine.train_weights_device()
        chk = torch.stack([v.double().sum() for v in flat.values()]).sum().reshape(1)
        allc = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(allc, chk)
        assert all(float(c) == float(allc[0]) for c in allc), "ranks diverged"      # same update everywhere
        if rank != 0:
            dist.barrier(); dist.destroy_process_group()
            return
        print("data parallel x%d (%s): weights identical on every rank after %d updates; gradient all-reduce %.1f ms"
              % (world, dist.get_backend(), n + 2, t_ar))
    print("config5 (fine-tune step, %s gradients, VGG16 + %s, B=%d, T=%d, %d heat-maps): %.1f ms/iteration; predict %.1f, lrp_weight %.1f, "
          "gradients %.1f, Adam + operand rebuild %.1f ms; losses %s; workspace %.1f GB" % (
              eng.train_precision, kind, B, T, n_maps, t_all / n, t_pred, t_lrp, t_step, t_apply, [round(float(v), 4) for v in losses.cpu()],
              eng.workspace_bytes / 1e9))
    assert torch.isfinite(g).all()
    if world > 1:
        dist.barrier(); dist.destroy_process_group()
        return


if __name__ == "__main__":
    main()
