#!/usr/bin/env python3
"""bench.py — LRP heat-maps/sec (per predicted token), VGG16 + adaptive attention, 224x224.

One "step" = one pass of the hot path over one batch of synthetic input that is
already resident in HBM: encode B images (CNN forward + relevance-gate caches),
replay the decoder for their captions, and produce one 224x224x3 relevance map
per predicted token (B x T heat-maps) through decoder-LRP -> CNN-LRP.
Workload (BASELINE.json configs[1]): B=32 images per GPU, T=10 words, V=10000.
Multi-GPU: one process per GPU, frozen weights broadcast from rank 0 over RCCL,
images sharded per rank, no data-path collective (weak scaling).

Prints ONE json line (rank 0).  See DESIGN.md §measurement for the roofline terms.
"""
import argparse
import json
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "LRP heatmaps/sec (per predicted token) VGG16+adaptive-attn, 224x224"
PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs @ 2.4 GHz
PEAK_BF16_MFMA_TFLOPS = 2500.0        # MI355X_MICROARCH.md: bf16 MFMA dense (v_mfma_f32_32x32x16_bf16)


def synth_weights(seed, V):
    from lrp_imagecaptioning_amd.synthetic import adaptive_weights, vgg_weights
    rs = np.random.RandomState(seed)
    w = vgg_weights(rs)
    w.update(adaptive_weights(rs, 196, 512, 512, 512, V))
    return w


def cpu_baseline(w, V, T, sample_tokens):
    """The oracle ("port" of the reference algorithm, literal cost structure) timed on the
    host cores: one image, `sample_tokens` of its T tokens, float32 CNN graph like TF."""
    import torch
    from lrp_imagecaptioning_amd.synthetic import VGG16_CFG, captions, images
    from oracle import cnn_lrp_ref as C
    from oracle.decoder_ref import AdaptiveOracle
    rs = np.random.RandomState(123)
    X = images(rs, 1)
    cap = captions(rs, 1, T, V)[0]
    layers = C.vgg_layers(w, VGG16_CFG)
    t0 = time.time()
    feat = C.forward(layers, X, torch.float32)                       # _image_model.predict (E:375)
    o = AdaptiveOracle(w, 196, 512, 512, 512)
    o.forward(feat, cap)                                             # _forward_beam_search
    t_fwd = time.time() - t0
    t1 = time.time()
    for t in range(1, sample_tokens + 1):
        R, _ = o.explain(t)                                          # _explain_lstm_single_word_sequence
        C.analyze(layers, X, R, torch.float32)                       # _explain_CNN (full graph per token, AB:511)
    t_tok = (time.time() - t1) / sample_tokens
    per_heatmap = t_tok + t_fwd / T                                  # forward replay amortised over the caption
    return {"value": round(1.0 / per_heatmap, 4), "unit": "heatmaps/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": "1 image, %d of %d tokens: literal rule-per-call decoder (numpy) + literal 5-pass "
                      "iNNvestigate graph per token (torch-CPU fp32); %.2f s/token + %.2f s forward/caption"
                      % (sample_tokens, T, t_tok, t_fwd)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU")
    ap.add_argument("--tokens", type=int, default=10, help="predicted words per caption")
    ap.add_argument("--vocab", type=int, default=10000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-tokens", type=int, default=10, help="tokens of one caption the CPU port explains (~1 s each)")
    ap.add_argument("--precision", default="bf16x3", choices=["bf16x3", "fp32", "bf16x3_fast"],
                    help="arithmetic of the per-token reverse walk: split-bf16 x3 MFMA (default) or exact fp32 MFMA")
    ap.add_argument("--handles", type=int, default=2,
                    help="batches in flight per GPU: consecutive steps alternate between this many lrp_handles on their own "
                         "HIP streams (pipeline.py); 1 = strictly one step after the other")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse "
                                                        "the N>1 control path with several ranks on one GPU)")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    local = local % max(torch.cuda.device_count(), 1)      # (rehearsal: more ranks than GPUs share a device)
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    from lrp_imagecaptioning_amd.parallel import broadcast_weights, shard_range
    from lrp_imagecaptioning_amd.pipeline import LRPPipeline
    from lrp_imagecaptioning_amd.synthetic import captions, images

    B, T, V = args.batch, args.tokens, args.vocab
    pipe = LRPPipeline(max(1, args.handles), decoder="adaptive", V=V, max_images=B, max_tokens=B * T, max_caption_len=T + 1,
                       device=local)
    eng = pipe.engines[0]
    pipe.set_precision(args.precision)
    # frozen weights: rank 0 owns them, everyone else receives them over RCCL/xGMI
    w_host = synth_weights(0, V) if rank == 0 else None
    if world > 1:
        pipe.set_weights_from_device(broadcast_weights(w_host, synth_weights_shapes(V), local, dist))
    else:
        pipe.set_weights(w_host)

    # this rank's shard of the global batch (global batch = world * B images)
    lo, hi = shard_range(world * B, world, rank)
    rs = np.random.RandomState(1000 + rank)
    X = torch.as_tensor(images(rs, hi - lo)).cuda(local)
    caps = captions(rs, hi - lo, T, V)
    img_idx = [b for b in range(B) for _ in range(T)]
    tpos = [t for _ in range(B) for t in range(1, T + 1)]
    outs = [torch.empty((B * T, 224, 224, 3), dtype=torch.float32, device=X.device) for _ in pipe.engines]
    out = outs[0]

    def step():
        # one pass of the hot path over one batch; consecutive steps go to alternating handles / streams
        pipe.explain_batch(X, caps, img_idx, tpos, out=outs[pipe.next_slot])

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=X.device if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert all(bool(torch.isfinite(o).all()) for o in outs)

    # dominant kernel, measured live with HIP events on the launch stream (outside the timed region, nothing else in flight)
    pipe.reset()
    eng.profile_enable(True)
    step()
    torch.cuda.synchronize()
    n_launch, ms, flop = eng.profile_query()
    eng.profile_enable(False)
    achieved = flop / (ms * 1e-3) / 1e12 if ms > 0 else 0.0

    split = args.precision != "fp32"
    traffic, traffic_src = pmc_traffic_per_launch("bf16x3" if split else "fp32")
    if rank == 0:
        heatmaps = world * B * T * args.steps
        if split:
            peak, dtype = PEAK_BF16_MFMA_TFLOPS, "bf16x3"
            kname = ("conv_igemm_kernel<..., PREC_BF16X3> (conv-LRP alpha1beta0 backward, 13 launches/step; every fp32 "
                     "product = 3 bf16 MFMAs hi*hi+hi*lo+lo*hi, fp32 accumulate; forward/decoder stay fp32/fp64)")
        else:
            peak, dtype = PEAK_F32_MFMA_TFLOPS, "f32"
            kname = "conv_igemm_kernel<..., PREC_FP32> (conv-LRP alpha1beta0 backward, 13 launches/step)"
        roof = {"bound": "mfma", "kernel": kname, "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4), "traffic": traffic,
                "traffic_unit": "HBM+MALL bytes per launch (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE)",
                "traffic_source": traffic_src, "launches": n_launch, "avg_launch_ms": round(ms / max(n_launch, 1), 4),
                "algorithmic_gflop_per_launch": round(flop / max(n_launch, 1) / 1e9, 2)}
        if split:
            roof["mfma_flop_per_algorithmic_flop"] = 3
            roof["issued_mfma_frac"] = round(3 * achieved / peak, 4)
            roof["vs_fp32_mfma_peak"] = round(achieved / PEAK_F32_MFMA_TFLOPS, 3)
        res = {
            "metric": METRIC, "value": round(heatmaps / dt, 2), "unit": "heatmaps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": "batch=%d synthetic 224x224 per GPU, VGG16 + adaptive-attention, LRP per-token "
                                   "heat-maps, %d words/caption, V=%d (BASELINE configs[1])" % (B, T, V),
                       "heatmaps_per_step": world * B * T, "parallelism": "image-sharded x%d, RCCL weight broadcast" % world,
                       "reverse_walk_precision": args.precision,
                       "handles_per_gpu": len(pipe.engines),
                       "schedule": "consecutive steps alternate between the handles, each on its own HIP stream (one batch's "
                                   "decoder / encode phases run under the previous batch's reverse walk)"
                                   if len(pipe.engines) > 1 else "one step after the other on one stream"},
            "roofline": roof,
        }
        if not args.no_cpu_baseline and args.cpu_sample_tokens > 0 and world == 1:               # (rank 0 at N = 1 only: the other ranks would idle behind it)
            res["cpu_baseline"] = cpu_baseline(w_host, V, T, args.cpu_sample_tokens)
        print(json.dumps(res))
    if dist:
        dist.barrier()
        dist.destroy_process_group()


def pmc_traffic_per_launch(precision="bf16x3"):
    """rocprofv3 cannot run inside this process: the per-launch fabric traffic of the reverse-walk
    conv launches comes from the newest committed PMC summary (profiles/run_profile.sh: separate
    --pmc FETCH_SIZE / WRITE_SIZE passes of this same command; profiles/summarize.py)."""
    import glob
    # the committed summaries are of the default-precision run; an fp32-mode profile would be r*_pmc_summary_fp32.json
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary%s.json" % ("" if precision == "bf16x3" else "_fp32"))))
    if not files:
        return None, None
    d = json.load(open(files[-1]))
    tot = n = 0.0
    for k, v in d.items():
        # reverse-walk launches: epilogues MUL(2) / MUL_UP2(3) / STORE(5) / fused image layer (6)
        m = re.search(r"conv_igemm_kernel<\d+, \d+, \d+, \d+, (\d+), (\d+)(?:, \w+)*>", k)
        if m and m.group(1) in ("2", "3", "5", "6") and m.group(2) == ("1" if precision == "bf16x3" else "0"):
            if "fetch_bytes_per_launch_x2corr" in v and "write_bytes_per_launch" in v:
                tot += (v["fetch_bytes_per_launch_x2corr"] + v["write_bytes_per_launch"]) * v["launches"]
                n += v["launches"]
    return (int(tot / n), os.path.relpath(files[-1], ROOT)) if n else (None, None)


def synth_weights_shapes(V):
    """name -> shape, identical on every rank (needed to size the broadcast buffer)."""
    from lrp_imagecaptioning_amd.synthetic import VGG16_CFG
    shp = {}
    for name, cin, cout, _ in VGG16_CFG:
        shp[name + "_W"] = (3, 3, cin, cout)
        shp[name + "_b"] = (cout,)
    H = E = D = 512
    shp.update({"image_features_W": (D, H), "image_features_b": (H,), "global_W": (D, E), "global_b": (E,),
                "embedding": (V, E), "lstm_Wi": (2 * E, 4 * H), "lstm_Wh": (H, 4 * H), "lstm_b": (4 * H,),
                "Wv": (H, H), "Wg": (H, H), "V": (H, 1), "Wx": (2 * E, H), "Wh": (H, H), "Ws": (H, H),
                "output_W": (H, V), "output_b": (V,)})
    return shp


if __name__ == "__main__":
    main()
