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

The line is self-certifying (N = 1): besides the timed default mode (split-bf16 reverse walk: hi*hi' + hi*lo' + lo*hi' in
every layer, 16 mantissa bits on both operands — parity independent of the weight statistics) the same invocation
  * times the exact-fp32 mode on the same batch              -> "fp32_mode"  {value, ms_per_step, roofline}
  * times the opt-in two-MFMA fp16 walk on the same batch    -> "fast_mode"  {value, ms_per_step, parity}
  * times the opt-in 2:4-sparse pooled boundaries            -> "sparse_pool_mode" {value, ms_per_step, distance from the default path}
    (one fp16 per weight below the top block: faster, but its error depends on the weights; lrp_hip.h)
  * checks sampled heat-maps of the timed batch, in every mode, against the CPU oracle
    (outside the timed region)                               -> "parity"     {bf16x3, fp32, f16x2: worst relative L1}
  * explains ONE image's 10-word caption on a B = 1 handle   -> "latency"    {ms, launches}  (the reference's own call,
    explain_image.py:152-161)
  * reads board power and shader clock while the steps run   -> "power"      {socket_power_w, sclk_mhz}
  * measures the fabric traffic of the dominant kernel live: two `rocprofv3 --pmc` child passes of this same
    script (FETCH_SIZE, WRITE_SIZE; started BEFORE this process touches the GPU) -> roofline.traffic
"""
import argparse
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
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


PARITY_SAMPLES = [(0, 1), (0, 10), (31, 10), (13, 5)]        # (image, t) of the timed batch checked against the oracle


def algorithmic_bytes_per_walk(n_tokens, n_images, dataflow="as_built"):
    """Bytes one reverse walk has to move, 4 B per element (split8 pairs are the same bytes as fp32), per launch
    S_in + S_out once each and the gate of every DISTINCT image once (the tokens of an image share it).
      "survey"    SURVEY 8d / rounds 1-3: every tensor at the resolution of the layer that reads it — S through a 2x2 pool
                  counted 4x-expanded, S_1 (n x 224^2 x 64) written by block1_conv2 and read by the image layer.  Still
                  the dataflow of the fp32 and f16x2 modes.
      "as_built"  the default (bf16x3) walk since round 3: a tensor that crosses a pool is stored ONCE at
                  POOLED resolution (the producer multiplies with the compact gate: value per window and channel, read by
                  the producer; the consumer reads one position byte per window and channel), and S_1 is not stored at
                  all — block1_conv2's launch carries the image layer and writes R_img (n x 224^2 x 3) from x (per image).
                  Partial sums, halos and re-fetches are NOT in this figure: traffic / this = the waste.
    Returns (total bytes, launches)."""
    from lrp_imagecaptioning_amd.synthetic import VGG16_CFG
    cfg = VGG16_CFG
    res, r = [], 224
    for _, cin, cout, pool in cfg:
        res.append(r)
        r = r // 2 if pool else r
    tot = 0
    for li, (_, cin, cout, pool) in enumerate(cfg):
        r_here = res[li]
        if dataflow == "survey":
            s_in = n_tokens * r_here ** 2 * cout * 4
            r_out = res[li - 1] if li else res[0]
            tot += s_in + n_tokens * r_out ** 2 * cin * 4 + n_images * r_out ** 2 * cin * 4
            continue
        if li == 0:
            continue                                       # the image layer rides on block1_conv2's launch (below)
        # what this launch reads: S at its own resolution — or the pooled pairs + position bytes when a pool follows it
        if pool:
            s_in = n_tokens * (r_here // 2) ** 2 * cout * 4 + n_images * (r_here // 2) ** 2 * cout * 1
        else:
            s_in = n_tokens * r_here ** 2 * cout * 4
        # what it writes: S for the layer below at THIS resolution (x its dense gate, or x the compact gate value when the
        # layer below is pooled); block1_conv2 instead finishes the image layer: R_img from x and G_1
        if li == 1:
            s_out = n_tokens * r_here ** 2 * 3 * 4 + n_images * r_here ** 2 * (3 + cin) * 4
        else:
            s_out = n_tokens * r_here ** 2 * cin * 4 + n_images * r_here ** 2 * cin * 4
        tot += s_in + s_out
    return tot, len(cfg)


def power_probe(step, torch, seconds=2.0):
    """rocm-smi readings (socket power, sclk) taken while a helper thread keeps issuing the bench's steps: an untimed
    look at the operating point the timed number was measured at.  None when rocm-smi is not there."""
    import re
    import shutil
    import subprocess
    import threading
    smi = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    if not os.path.exists(smi):
        return None
    stop = threading.Event()

    def work():
        k = 0
        while not stop.is_set():
            step()
            k += 1
            if k % 4 == 0:                                # (bounded queue; the handles keep overlapping as in the timed run)
                torch.cuda.synchronize()
        torch.cuda.synchronize()
    th = threading.Thread(target=work)
    th.start()
    watts, mhz = [], []
    t0 = time.perf_counter()
    try:
        time.sleep(0.5)
        while time.perf_counter() - t0 < seconds:
            out = subprocess.run([smi, "-d", str(torch.cuda.current_device()), "--showpower", "--showclocks"],
                                 capture_output=True, text=True, timeout=20).stdout
            m = re.search(r"Power \(W\):\s*([0-9.]+)", out)
            c = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", out)
            if m:
                watts.append(float(m.group(1)))
            if c:
                mhz.append(int(c.group(1)))
    except Exception:
        pass
    finally:
        stop.set()
        th.join()
    if not watts or not mhz:
        return None
    return {"socket_power_w": round(float(np.median(watts)), 1), "sclk_mhz": int(np.median(mhz)), "samples": len(watts),
            "source": "rocm-smi --showpower --showclocks while the bench's steps run (untimed)"}


def sustained_probe(step, torch, seconds, est_ms_per_step, heatmaps_per_step):
    """The same steps back to back for >= `seconds` (and >= 150 steps): the headline's timed region is ~0.6 s on a board
    that sits at its power limit, so this is the rate it HOLDS — with socket power and shader clock sampled by rocm-smi
    over the same window (a sampler thread; the main thread issues the steps exactly as the timed run does)."""
    import math
    import threading
    n = max(150, int(math.ceil(seconds * 1e3 / max(est_ms_per_step, 1e-3))))
    smi = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    watts, mhz = [], []
    stop = threading.Event()

    def sample():
        dev = str(torch.cuda.current_device())
        while not stop.is_set():
            try:
                out = subprocess.run([smi, "-d", dev, "--showpower", "--showclocks"], capture_output=True, text=True, timeout=20).stdout
            except Exception:
                return
            m = re.search(r"Power \(W\):\s*([0-9.]+)", out)
            c = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", out)
            if m:
                watts.append(float(m.group(1)))
            if c:
                mhz.append(int(c.group(1)))
    th = threading.Thread(target=sample) if os.path.exists(smi) else None
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    if th:
        th.start()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stop.set()
    if th:
        th.join()
    blk = {"value": round(heatmaps_per_step * n / dt, 2), "unit": "heatmaps/s", "steps": n, "seconds": round(dt, 2),
           "ms_per_step": round(dt / n * 1e3, 3),
           "what": "the timed workload, %d steps back to back without a host synchronise in between" % n}
    if watts and mhz:
        blk.update({"socket_power_w": round(float(np.median(watts)), 1), "socket_power_w_max": round(max(watts), 1),
                    "sclk_mhz": int(np.median(mhz)), "sclk_mhz_min": min(mhz), "samples": len(watts),
                    "source": "rocm-smi --showpower --showclocks sampled over the same window"})
    return blk


def top_block_flop_share():
    """share of the reverse walk's algorithmic flops in the layers after the last pool (the three-MFMA layers of f16x2)"""
    from lrp_imagecaptioning_amd.synthetic import VGG16_CFG
    r, fl, last_pool = 224, [], -1
    for li, (_, cin, cout, pool) in enumerate(VGG16_CFG):
        fl.append(r * r * 9.0 * cout * (6 if li == 0 else cin))
        if pool:
            last_pool, r = li, r // 2
    return sum(fl[last_pool + 1:]) / sum(fl)


def oracle_heatmaps(w, X, caps, samples):
    """Float64 oracle (oracle/: decoder pinned by the reference's own outputs, CNN = literal iNNvestigate graph) for
    the sampled (image, t) pairs of the timed batch: {(b, t): (224, 224, 3) relevance}."""
    from lrp_imagecaptioning_amd.synthetic import VGG16_CFG
    from oracle import cnn_lrp_ref as C
    from oracle.decoder_ref import AdaptiveOracle
    layers = C.vgg_layers(w, VGG16_CFG)
    refs, dec = {}, {}
    for (b, t) in samples:
        if b not in dec:
            feat = C.forward(layers, X[b:b + 1]).astype(np.float32)
            dec[b] = AdaptiveOracle(w, 196, 512, 512, 512)
            dec[b].forward(feat, caps[b])
        R, _ = dec[b].explain(t)
        refs[(b, t)] = C.analyze(layers, X[b:b + 1], R)[0]
    return refs


def rel_l1(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).sum() / np.abs(b).sum())


WALK_KERNEL_RE = re.compile(r"conv_igemm_kernel<\d+, \d+, \d+, \d+, (\d+), (\d+)(?:, \w+)*>")


PREC_TEMPLATE_ARG = {"fp32": "0", "bf16x3": "1", "bf16x3_fast": "1", "f16x2": "2"}   # conv_igemm.h ConvPrec


def is_walk_kernel(name, precision):
    """reverse-walk launches: epilogues MUL(2) / MUL_UP2(3) / STORE(5) / fused image layer (6) in the walk's arithmetic"""
    if "img_partial_sum_kernel" in name:                       # second half of the image layer when it is folded into block1_conv2
        return precision in ("bf16x3", "bf16x3_fast")
    m = WALK_KERNEL_RE.search(name)
    return bool(m and m.group(1) in ("2", "3", "5", "6") and m.group(2) == PREC_TEMPLATE_ARG[precision])


def live_pmc_traffic(args, precision=None):
    """HBM/fabric bytes per reverse-walk launch, measured in THIS run: two rocprofv3 child passes (FETCH_SIZE and
    WRITE_SIZE need separate passes: TCC has 4 counter slots, MI355X_MICROARCH.md) of `bench.py --child` = one step of
    the same workload.  FETCH_SIZE is doubled (gfx950 tallies the 128 B requests of 16 B/lane streams at 64 B).  Must
    run before this process initialises the GPU (children only; nothing here is exec'ed over a GPU process)."""
    import csv
    import glob
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not found"
    precision = precision or args.precision
    tot = {}
    n = {}
    tmp = tempfile.mkdtemp(prefix="lrp_pmc_")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, ctr)
            cmd = [exe, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", out, "-o", "p", "--",
                   sys.executable, os.path.abspath(__file__), "--child", "--steps", "1", "--warmup", "0",
                   "--batch", str(args.batch), "--tokens", str(args.tokens), "--vocab", str(args.vocab),
                   "--precision", precision, "--handles", str(args.handles)]
            env = dict(os.environ, TMPDIR="/tmp")
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=420)
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, "rocprofv3 --pmc %s pass failed (rc %d)" % (ctr, r.returncode)
            seen = set()
            tot[ctr] = 0.0
            for row in csv.DictReader(open(files[0])):
                if row["Counter_Name"] == ctr and is_walk_kernel(row["Kernel_Name"], precision):
                    tot[ctr] += float(row["Counter_Value"])
                    seen.add(row["Dispatch_Id"])
            n[ctr] = len(seen)
            if not n[ctr]:
                return None, "no reverse-walk dispatches in the %s pass" % ctr
    except Exception as e:                                     # (a profiler that cannot run must not cost the bench line)
        return None, "live PMC passes failed: %s" % e
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    fetch = 2.0 * tot["FETCH_SIZE"] * 1024 / n["FETCH_SIZE"]
    write = tot["WRITE_SIZE"] * 1024 / n["WRITE_SIZE"]
    return {"bytes": int(fetch + write), "fetch_x2": int(fetch), "write": int(write), "launches": n["FETCH_SIZE"]}, \
        "live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes of this invocation (one step each)"


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
    ap.add_argument("--precision", default="bf16x3", choices=["bf16x3", "fp32", "f16x2", "bf16x3_fast"],
                    help="arithmetic of the per-token reverse walk: split-bf16 x3 MFMA (library default), exact fp32 MFMA, or "
                         "the opt-in fast mode: fp16 pairs with ONE fp16 per weight below the top block (2 MFMAs per product)")
    ap.add_argument("--no-power", action="store_true", help="skip the rocm-smi power / clock reading")
    ap.add_argument("--handles", type=int, default=2,
                    help="batches in flight per GPU: consecutive steps alternate between this many lrp_handles on their own "
                         "HIP streams (pipeline.py); 1 = strictly one step after the other")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse "
                                                        "the N>1 control path with several ranks on one GPU)")
    ap.add_argument("--no-fp32-mode", action="store_true", help="skip the exact-fp32 and fast-mode blocks of the line")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-image latency block")
    ap.add_argument("--no-parity", action="store_true", help="skip the in-run oracle check of sampled heat-maps")
    ap.add_argument("--no-pmc", action="store_true", help="skip the live rocprofv3 traffic passes (use the committed summary)")
    ap.add_argument("--sustained-seconds", type=float, default=5.0,
                    help="length of the `sustained` block: the same steps back to back for at least this long (0 = skip)")
    ap.add_argument("--no-configs", action="store_true", help="skip the config4 / config5 sub-blocks (BASELINE configs[3], [4])")
    ap.add_argument("--child", action="store_true", help=argparse.SUPPRESS)    # one plain step under rocprofv3 (live_pmc_traffic)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    extras = world == 1 and not args.child                  # the self-certifying blocks: rank 0 of a one-GPU run only
    live_traffic, live_src = (None, None)
    live32, live32_src = (None, None)
    if extras and not args.no_pmc:
        live_traffic, live_src = live_pmc_traffic(args)     # children first: this process has not touched the GPU yet
        if not args.no_fp32_mode and args.precision != "fp32":
            live32, live32_src = live_pmc_traffic(args, "fp32")

    import torch
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
    bcast = None
    if world > 1:
        shapes = synth_weights_shapes(V)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        bundle = broadcast_weights(w_host, shapes, local, dist)
        torch.cuda.synchronize()
        bcast = {"bytes": 4 * sum(int(np.prod(sh)) for sh in shapes.values()), "ms": round((time.perf_counter() - t0) * 1e3, 2),
                 "what": "ONE dist.broadcast of the flat frozen bundle from rank 0 (host flatten + H2D on rank 0 included), "
                         "then lrp_set_weight_dev packs it on each GPU"}
        pipe.set_weights_from_device(bundle)
    else:
        pipe.set_weights(w_host)

    # this rank's shard of the global batch (global batch = world * B images)
    lo, hi = shard_range(world * B, world, rank)
    rs = np.random.RandomState(1000 + rank)
    X_host = images(rs, hi - lo)
    X = torch.as_tensor(X_host).cuda(local)
    caps = captions(rs, hi - lo, T, V)
    img_idx = [b for b in range(B) for _ in range(T)]
    tpos = [t for _ in range(B) for t in range(1, T + 1)]
    outs = [torch.empty((B * T, 224, 224, 3), dtype=torch.float32, device=X.device) for _ in pipe.engines]

    def step():
        # one pass of the hot path over one batch; consecutive steps go to alternating handles / streams
        pipe.explain_batch(X, caps, img_idx, tpos, out=outs[pipe.next_slot])

    per_rank = []                                           # seconds of the last timed_run on every rank

    def timed_run(n_warm, n_steps):
        """W untimed + exactly K timed steps, barrier + device synchronise on both sides, MAX over ranks."""
        for _ in range(n_warm):
            step()
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            step()
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        per_rank[:] = [dt]
        if dist:
            cdev = X.device if args.backend == "nccl" else "cpu"
            tt = torch.tensor([dt], dtype=torch.float64, device=cdev)
            every = [torch.zeros_like(tt) for _ in range(world)]
            dist.all_gather(every, tt)                        # each rank's own clock, for the line's `distributed` block
            per_rank[:] = [float(e.item()) for e in every]
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        assert all(bool(torch.isfinite(o).all()) for o in outs)
        return dt

    def dominant_kernel(reps=3):
        """reverse-walk launches of `reps` whole steps on handle 0, HIP events on the launch stream (one handle alone: nothing
        else in flight); returns launches per step and the per-step averages of their summed time and algorithmic FLOPs"""
        eng.profile_enable(True)
        for _ in range(reps):
            pipe.reset()                                        # (every profiled step goes to handle 0)
            step()
            torch.cuda.synchronize()
        n_launch, ms, flop = eng.profile_query()
        eng.profile_enable(False)
        pipe.reset()
        return n_launch // reps, ms / reps, flop / reps

    def roofline_block(precision, n_launch, ms, flop, traffic, traffic_src):
        split = precision != "fp32"
        achieved = flop / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        per_product = 3
        if precision == "f16x2":
            peak = PEAK_BF16_MFMA_TFLOPS                     # (v_mfma_f32_32x32x16_f16 runs at the bf16 rate)
            per_product = round(2.0 + top_block_flop_share(), 3)   # 3 MFMAs per product after the last pool, 2 below
            kname = ("conv_igemm_kernel<..., PREC_F16X2> (conv-LRP alpha1beta0 backward, 13 launches/step; relevance as an "
                     "fp16 pair hi+lo with per-token power-of-two scales, weights as scaled fp16 pairs: 3 f16 MFMAs per "
                     "product in block5, 2 (weights' hi half only) below, fp32 accumulate; forward/decoder stay "
                     "fp32-grade/fp64)")
        elif split:
            peak = PEAK_BF16_MFMA_TFLOPS
            kname = ("conv_igemm_kernel<..., PREC_BF16X3> (conv-LRP alpha1beta0 backward, 13 launches/step; every fp32 "
                     "product = 3 bf16 MFMAs hi*hi+hi*lo+lo*hi, fp32 accumulate; forward/decoder stay fp32/fp64)")
        else:
            peak = PEAK_F32_MFMA_TFLOPS
            kname = "conv_igemm_kernel<..., PREC_FP32> (conv-LRP alpha1beta0 backward, 13 launches/step)"
        # the default walk moves the compact dataflow; fp32 / f16x2 walks still move the expanded one
        flow = "as_built" if precision in ("bf16x3", "bf16x3_fast") else "survey"
        abytes, alaunch = algorithmic_bytes_per_walk(B * T, B, flow)
        roof = {"bound": "mfma", "kernel": kname, "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4), "traffic": traffic,
                "traffic_unit": "HBM+MALL bytes per launch (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE)",
                "traffic_source": traffic_src, "launches": n_launch, "avg_launch_ms": round(ms / max(n_launch, 1), 4),
                "algorithmic_gflop_per_launch": round(flop / max(n_launch, 1) / 1e9, 2),
                "algorithmic_bytes_per_launch": int(abytes / alaunch), "algorithmic_bytes_dataflow": flow,
                "algorithmic_bytes_per_launch_survey8d": int(algorithmic_bytes_per_walk(B * T, B, "survey")[0] / alaunch),
                "traffic_over_algorithmic_bytes": round(traffic / (abytes / alaunch), 3) if traffic else None}
        if split:
            roof["mfma_flop_per_algorithmic_flop"] = per_product
            roof["issued_mfma_frac"] = round(per_product * achieved / peak, 4)
            roof["vs_fp32_mfma_peak"] = round(achieved / PEAK_F32_MFMA_TFLOPS, 3)
        return roof

    def sampled(precision):
        """the sampled heat-maps of the timed batch as the last step on handle 0 left them"""
        return {(b, t): outs[0][b * T + t - 1].cpu().numpy() for (b, t) in PARITY_SAMPLES if b < B and t <= T}

    # ---- the timed region (default arithmetic)
    dt = timed_run(args.warmup, args.steps)
    if args.child:
        return
    per_rank_ms = [round(t / args.steps * 1e3, 3) for t in per_rank]
    n_launch, ms, flop = dominant_kernel()
    got = {args.precision: sampled(args.precision)} if extras and not args.no_parity else {}

    # ---- the rate the board HOLDS: the same steps for >= 5 s, power and clock sampled over that window
    sustained_block = None
    if extras and args.sustained_seconds > 0:
        pipe.reset()
        sustained_block = sustained_probe(step, torch, args.sustained_seconds, dt / args.steps * 1e3, B * T)
        assert all(bool(torch.isfinite(o).all()) for o in outs)
        pipe.reset()

    # ---- board power / shader clock while the same steps run (untimed): the walk is power-limited on MI355X, which is
    # why its MFMA count, not the overlap of its phases, sets the rate
    power_block = power_probe(step, torch) if extras and not args.no_power else None

    # ---- the same batch in exact-fp32 arithmetic (the reference's: TF float32 / numpy float64), same invocation
    fp32_block = None
    if extras and not args.no_fp32_mode and args.precision != "fp32":
        pipe.reset()
        pipe.set_precision("fp32")
        k32 = max(2, min(args.steps, 10))
        dt32 = timed_run(1, k32)
        nl32, ms32, fl32 = dominant_kernel()
        if not args.no_parity:
            got["fp32"] = sampled("fp32")
        fp32_block = {"value": round(B * T * k32 / dt32, 2), "unit": "heatmaps/s", "steps": k32, "warmup": 1,
                      "ms_per_step": round(dt32 / k32 * 1e3, 3), "dtype": "f32",
                      "roofline": roofline_block("fp32", nl32, ms32, fl32,
                                                 *((live32["bytes"], live32_src) if live32 else pmc_traffic_per_launch("fp32")))}
        if live32:
            fp32_block["roofline"]["traffic_detail"] = live32
        elif live32_src:
            fp32_block["roofline"]["traffic_source"] = live32_src
        pipe.reset()
        pipe.set_precision(args.precision)

    # ---- and in the opt-in fast mode (fp16 pairs, one fp16 per weight below the top block: two MFMAs per product): what it
    # would buy on THESE weights and what it costs in parity on them — its error depends on the weight statistics (dense
    # Gaussian kernels average the per-weight rounding out; sparse heavy-tailed ones do not: tests/test_gpu_stress_parity.py)
    fast_block = None
    if extras and not args.no_fp32_mode and args.precision == "bf16x3":
        from lrp_imagecaptioning_amd.calibration import calibrate_fast_mode
        pipe.reset()
        # which layers may take the two-MFMA form is MEASURED on these weights first (calibration.py: per-layer heat-map
        # error against the exact-fp32 mode on two images of the batch, 10x margin under the 1e-4 bar), then timed
        cal = calibrate_fast_mode(eng, X[:2], tolerance=1e-4, margin=10.0, apply=False)
        pipe.set_precision("f16x2")
        pipe.set_fast_layers(cal["mask"])
        kb = max(2, min(args.steps, 10))
        dtb = timed_run(2, kb)
        if not args.no_parity:
            got["f16x2"] = sampled("f16x2")
        fast_block = {"value": round(B * T * kb / dtb, 2), "unit": "heatmaps/s", "steps": kb, "warmup": 2,
                      "ms_per_step": round(dtb / kb * 1e3, 3), "dtype": "f16x2",
                      "two_term_layers": cal["layers"],
                      "calibration": {"reference": "exact-fp32 mode, same images and relevances (CNN half)",
                                      "budget": cal["budget"], "error_of_the_mix": cal["error"], "three_mfma_floor": cal["floor"],
                                      "per_layer_error": {k: float("%.3g" % v) for k, v in cal["per_layer"].items()},
                                      "images": cal["n_images"], "relevance_maps": cal["n_relevances"]},
                      "note": "opt-in lrp_set_precision(LRP_PREC_F16X2) + lrp_set_fast_layers(calibrated mask): ONE fp16 per "
                              "weight (11 bits) in the layers listed, fp16 pairs x fp16 pairs (three MFMAs) elsewhere; the "
                              "mask and the parity below are for THESE synthetic He-normal weights — on sparse heavy-tailed "
                              "kernels the same calibration qualifies no layer (tests/test_gpu_calibration.py)"}
        pipe.reset()
        pipe.set_fast_layers(None)
        pipe.set_precision(args.precision)

    # ---- and with the two >= 256-column pooled boundaries of the walk on the 2:4-sparse matrix cores (opt-in LRP_SPARSE_POOL=1,
    # below the 25 % per-launch gain that would make it the default, and it would end B = 1 == B = 32 to the bit)
    sparse_block = None
    if extras and not args.no_fp32_mode and args.precision == "bf16x3":
        from lrp_imagecaptioning_amd.engine import switches
        pipe.reset()
        with switches(LRP_SPARSE_POOL=1):
            ksp = max(2, min(args.steps, 10))
            dtsp = timed_run(2, ksp)
            sp_maps = sampled("bf16x3")
        sparse_block = {"value": round(B * T * ksp / dtsp, 2), "unit": "heatmaps/s", "steps": ksp, "warmup": 2,
                        "ms_per_step": round(dtsp / ksp * 1e3, 3), "dtype": "bf16x3",
                        "what": "LRP_SPARSE_POOL=1: block4_conv3 / block3_conv3 of the walk as v_smfmac_f32_32x32x32_bf16 launches "
                                "(12 slots per channel instead of 9 dense taps x 4 window positions)"}
        if got.get(args.precision):
            ref_maps = got[args.precision]
            sparse_block["rel_l1_vs_default_path"] = float(max(
                np.abs(sp_maps[k].astype(np.float64) - ref_maps[k]).sum() / np.abs(ref_maps[k].astype(np.float64)).sum() for k in ref_maps))
        pipe.reset()

    # ---- the reference's actual call: ONE image, explain every word of its caption (explain_image.py:152-161 ->
    # E:183-189): host clock around encode -> decoder replay -> T heat-maps -> device synchronise, on a B = 1 handle
    latency_block = None
    if extras and not args.no_latency:
        X1, what1 = real_image(torch, X.device)
        latency_block = latency_probe(w_host, X1 if X1 is not None else X[:1], caps[:1], T, V, local, args.precision, torch,
                                      image=what1 or "synthetic U[0,255] image")

    # ---- BASELINE configs[3] and [4] at one GPU's share, one timed run each with a sampled parity check, so that those
    # configurations have a driver-timed number too (the headline stays configs[1])
    config4_block = config5_block = None
    if extras and not args.no_configs:
        pipe.reset()
        for e in pipe.engines:
            e.close()
        del outs[:]
        torch.cuda.empty_cache()
        config4_block = guarded_block(config4_probe, torch, local, not args.no_parity)
        torch.cuda.empty_cache()
        config5_block = guarded_block(config5_probe, torch, local, not args.no_parity)

    if live_traffic:
        traffic, traffic_src = live_traffic["bytes"], live_src
    else:
        traffic, traffic_src = pmc_traffic_per_launch(args.precision)
        if traffic_src and live_src:
            traffic_src += " (%s)" % live_src
    if rank == 0:
        heatmaps = world * B * T * args.steps
        dtype = {"fp32": "f32", "f16x2": "f16x2"}.get(args.precision, "bf16x3")   # the arithmetic type the walk computes in
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
            "roofline": roofline_block(args.precision, n_launch, ms, flop, traffic, traffic_src),
        }
        if live_traffic:
            res["roofline"]["traffic_detail"] = live_traffic
        if fp32_block:
            res["fp32_mode"] = fp32_block
        if fast_block:
            res["fast_mode"] = fast_block
        if sparse_block:
            res["sparse_pool_mode"] = sparse_block
        if sustained_block:
            res["sustained"] = sustained_block
        if world > 1:
            res["distributed"] = {
                "backend": dist.get_backend(), "world_size": dist.get_world_size(),
                "nccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if args.backend == "nccl" else None,
                "ms_per_step_per_rank": per_rank_ms, "timing": "MAX over ranks (all_reduce) of each rank's own clock around the "
                "barrier-bracketed timed region", "weight_broadcast": bcast,
                "data_path_collectives_per_step": 0}
        if latency_block:
            res["latency"] = latency_block
        if config4_block:
            res["config4"] = config4_block
        if config5_block:
            res["config5"] = config5_block
        if power_block:
            res["power"] = power_block
        if got:
            # worst relative L1 (BASELINE: sum|R - R_ref| / sum|R_ref| on the raw (224,224,3) relevance, per token) of the
            # sampled heat-maps of the TIMED batch against the float64 oracle; outside the timed region
            t0 = time.time()
            refs = oracle_heatmaps(w_host, X_host, caps, list(next(iter(got.values())).keys()))
            par = {"metric": "max over samples of sum|R-R_ref|/sum|R_ref|", "tolerance": 1e-4,
                   "samples": [list(k) for k in refs], "oracle": "oracle/decoder_ref.py + oracle/cnn_lrp_ref.py (float64)",
                   "oracle_seconds": None}
            for mode, maps in got.items():
                par[mode] = max(rel_l1(maps[k], refs[k]) for k in refs)
            par["oracle_seconds"] = round(time.time() - t0, 1)
            # the verdict is about the timed arithmetic and the exact one; the opt-in fast mode is reported beside them
            par["ok"] = all(par[m] < par["tolerance"] for m in got if m != "f16x2" or args.precision == "f16x2")
            res["parity"] = par
            if fast_block and "f16x2" in par:
                fast_block["parity"] = par["f16x2"]
        if not args.no_cpu_baseline and args.cpu_sample_tokens > 0:
            if world == 1:                                       # (rank 0 at N = 1 only: the other ranks would idle behind it)
                res["cpu_baseline"] = cpu_baseline(w_host, V, T, args.cpu_sample_tokens)
            else:
                res["cpu_baseline"] = None                       # reported by the N = 1 line
        print(json.dumps(res))
    if dist:
        dist.barrier()
        dist.destroy_process_group()


def real_image(torch, device):
    """The reference's own example photograph (example_images/flickr30kimage/1009434119.jpg, explain_image.py:321-371;
    decoded pixels in tests/golden/real_images.npz), preprocessed on the device (models/preprocessors.py:38-53)."""
    f = os.path.join(ROOT, "tests", "golden", "real_images.npz")
    if not os.path.exists(f):
        return None, None
    from lrp_imagecaptioning_amd.engine import preprocess_images
    d = np.load(f)
    x = preprocess_images(torch.as_tensor(d["rgb_u8"][:1]).to(device))
    return x, "the reference's example photograph %s (tests/golden/real_images.npz), preprocessed on the device" % str(d["names"][0])


def guarded_block(fn, *a):
    """A sub-block that fails must not cost the headline line: its error is reported in its place."""
    try:
        return fn(*a)
    except Exception as e:                                  # noqa: BLE001
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}


def config4_probe(torch, device, parity=True, B=32, T=10, V=10000, steps=6):
    """BASELINE configs[3] at one GPU's share (batch = 128 on 4 GPUs): grid-TD decoder + ResNet-101 encoder, LRP alpha1beta0,
    32 images x 10 words per step, two batches in flight like the headline.  One timed run; one sampled heat-map against
    the float64 oracles (oracle/resnet_lrp_ref.py + GridTDOracle) outside the timed region."""
    from lrp_imagecaptioning_amd.pipeline import LRPPipeline
    from lrp_imagecaptioning_amd.synthetic import RESNET101_STACKS, captions, gridtd_weights, images, resnet_weights
    rs = np.random.RandomState(0)
    w = resnet_weights(rs)
    w.update(gridtd_weights(rs, 49, 2048, 512, 512, V))
    pipe = LRPPipeline(2, decoder="gridtd", img_hw=(224, 224), L=49, D=2048, H=512, E=512, V=V, max_images=B, max_tokens=B * T,
                       max_caption_len=T + 1, resnet={"stem": 64, "stacks": RESNET101_STACKS}, device=device)
    pipe.set_weights(w)
    X_host = images(rs, B)
    X = torch.as_tensor(X_host).cuda(device)
    caps = captions(rs, B, T, V)
    idx = [b for b in range(B) for _ in range(T)]
    tt = [t for _ in range(B) for t in range(1, T + 1)]
    outs = [torch.empty((B * T, 224, 224, 3), dtype=torch.float32, device=X.device) for _ in pipe.engines]
    step = lambda: pipe.explain_batch(X, caps, idx, tt, out=outs[pipe.next_slot])
    for _ in range(2):
        step()
    pipe.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert all(bool(torch.isfinite(o).all()) for o in outs)
    blk = {"value": round(B * T * steps / dt, 2), "unit": "heatmaps/s", "steps": steps, "warmup": 2,
           "ms_per_step": round(dt / steps * 1e3, 3), "dtype": "bf16x3", "workspace_gb": round(sum(e.workspace_bytes for e in pipe.engines) / 1e9, 1),
           "workload": "batch=%d synthetic 224x224 per GPU, ResNet-101 (conv5_block3_out) + grid-TD, LRP-alpha1beta0 per-token "
                       "heat-maps, %d words/caption, V=%d (BASELINE configs[3]: batch=128 on 4 GPUs)" % (B, T, V)}
    if parity:
        from oracle import resnet_lrp_ref as RN
        from oracle.decoder_ref import GridTDOracle
        t0 = time.time()
        b, t = 5, 2
        spec = RN.resnet_spec()
        o = GridTDOracle(w, 49, 2048, 512, 512)
        o.forward(RN.forward(w, spec, X_host[b:b + 1]).astype(np.float32), caps[b])
        ref = RN.analyze(w, spec, X_host[b:b + 1], o.explain(t)[0].reshape(1, 7, 7, 2048))[0]
        e = rel_l1(outs[0][b * T + t - 1].cpu().numpy(), ref)
        blk["parity"] = {"sample": [b, t], "rel_l1": e, "tolerance": 1e-4, "ok": bool(e < 1e-4),
                         "oracle": "oracle/resnet_lrp_ref.py + GridTDOracle (float64; unpinned: TensorFlow)", "oracle_seconds": round(time.time() - t0, 1)}
    for e in pipe.engines:
        e.close()
    return blk


def config5_probe(torch, device, parity=True, B=8, T=21, V=10000, iters=6):
    """BASELINE configs[4] at one GPU's share (batch = 64 on 8 GPUs): the LRP-inference fine-tune iteration (train.py:571-580)
    of the VGG16 + adaptive-attention captioner in bf16 gradient mode — predict, lrp_weight for every word of every
    predicted caption, gradients, Adam + operand rebuild.  One timed run; one sampled `lrp_weight` entry against the
    reference loop (M:1657-1689) on the float64 oracles."""
    from lrp_imagecaptioning_amd.explainers import CaptionModelSpec, ExplainImgCaptioningAdaptiveAttention
    from lrp_imagecaptioning_amd.synthetic import VGG16_CFG, adaptive_weights, images, vgg_weights
    from lrp_imagecaptioning_amd.training import TrainingLRPInferenceAdaptive
    torch.cuda.set_device(device)
    rs = np.random.RandomState(0)
    w = vgg_weights(rs)
    w.update(adaptive_weights(rs, 196, 512, 512, 512, V))
    spec = CaptionModelSpec(w, img_encoder="vgg16", hidden_dim=512, embedding_dim=512, L=196, D=512, vocab_size=V)
    ex = ExplainImgCaptioningAdaptiveAttention(spec, None, None, max_caption_length=T - 1, max_images=B)
    tr = TrainingLRPInferenceAdaptive(ex, learning_rate=2e-4, drop_rate=0.5)
    eng = ex._engine
    eng.train_set_precision("bf16")
    rs = np.random.RandomState(100)
    X_host = images(rs, B)
    X = torch.as_tensor(X_host).cuda(device)
    cap_in = np.concatenate([np.full((B, 1), 1), rs.randint(2, V, size=(B, T - 1))], axis=1).astype(np.int32)
    y = rs.randint(0, V, size=(B, T)).astype(np.int32)
    blk = {}
    if parity:                                              # before any update: the weights are still `w`
        from lrp_imagecaptioning_amd.postprocess import lrp_inference_score
        from oracle import cnn_lrp_ref as C
        from oracle.decoder_ref import AdaptiveOracle
        t0 = time.time()
        y_pred = tr.predict_on_batch([cap_in, X])
        lw = tr._lrp_layer.call_device(X, y_pred, images_encoded=True)
        words = (torch.argmax(y_pred, dim=-1) + 1).cpu().numpy()
        b, i = 3, 4
        cap = [int(c) for c in words[b]]
        full = cap[:cap.index(1) + 1] if 1 in cap else cap[:T - 1] + [1]
        if i < len(full) - 1:
            layers = C.vgg_layers(w, VGG16_CFG)
            o = AdaptiveOracle(w, 196, 512, 512, 512)
            o.forward(C.forward(layers, X_host[b:b + 1]).astype(np.float32), full)
            want = 1 + lrp_inference_score(C.analyze(layers, X_host[b:b + 1], o.explain(i + 1)[0]), "mean")
            got = float(lw[b, i, cap[i]])
            e = abs(got - want) / abs(want - 1)
            blk["parity"] = {"sample": [b, i], "lrp_weight_score_rel_err": e, "tolerance": 2e-3, "ok": bool(e < 2e-3),
                             "oracle": "reference loop M:1657-1689 on oracle/decoder_ref.py + oracle/cnn_lrp_ref.py (float64)",
                             "oracle_seconds": round(time.time() - t0, 1)}
    for _ in range(3):
        tr.train_on_batch([cap_in, X], y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        losses = tr.train_on_batch([cap_in, X], y)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n_maps = B * (T - 1)
    blk.update({"value": round(B * iters / dt, 2), "unit": "images/s", "iterations": iters, "warmup": 3,
                "ms_per_iteration": round(dt / iters * 1e3, 3), "heatmaps_per_iteration": n_maps, "dtype": "bf16 (conv weight gradients) / bf16x3 / fp32",
                "finite_losses": bool(np.all(np.isfinite(np.asarray([float(v) for v in losses])))),
                "workload": "LRP-inference fine-tune iteration, VGG16 + adaptive attention, batch=%d per GPU, T=%d, V=%d, every word of "
                            "every predicted caption explained (BASELINE configs[4]: batch=64 on 8 GPUs)" % (B, T, V)})
    eng.close()
    return blk


def latency_probe(w_host, X1, caps1, T, V, device, precision, torch, reps=7, image="synthetic"):
    """Single-image latency: a handle sized for ONE image and its T words (the small-tile kernels), weights resident,
    image resident in HBM; per repetition encode_images -> decoder_forward -> explain_tokens(T) -> synchronise, host clock.
    Reports the median, the best, and the kernel launches behind one repetition (lrp_launch_count)."""
    from lrp_imagecaptioning_amd import _capi
    from lrp_imagecaptioning_amd.engine import LRPEngine
    eng = LRPEngine(decoder="adaptive", V=V, max_images=1, max_tokens=T, max_caption_len=T + 1, device=device)
    eng.set_precision(precision)
    eng.set_weights(w_host)
    out = torch.empty((T, 224, 224, 3), dtype=torch.float32, device=X1.device)
    idx, tpos = [0] * T, list(range(1, T + 1))
    lib = _capi.load()
    ms, launches = [], None
    for r in range(reps + 2):
        torch.cuda.synchronize()
        n0 = int(lib.lrp_launch_count())
        t0 = time.perf_counter()
        eng.encode_images(X1)
        eng.decoder_forward(caps1)
        eng.explain_tokens(idx, tpos, out=out)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        if r >= 2:                                              # two warm-up repetitions (lazy allocations, code objects)
            ms.append(dt)
            launches = int(lib.lrp_launch_count()) - n0
    assert bool(torch.isfinite(out).all())
    eng.close()
    ms.sort()
    return {"ms": round(ms[len(ms) // 2], 3), "best_ms": round(ms[0], 3), "launches": launches, "heatmaps": T,
            "reps": reps, "dtype": {"fp32": "f32"}.get(precision, precision),
            "image": image,
            "what": "1 image resident in HBM, B = 1 handle: encode + decoder replay + %d per-word heat-maps + synchronise, "
                    "host clock (median of %d); explain_image.py:152-161" % (T, reps)}


def pmc_traffic_per_launch(precision="bf16x3"):
    """Fallback when the live rocprofv3 passes cannot run (no profiler, N > 1): the per-launch fabric traffic of the
    reverse-walk conv launches from the newest committed PMC summary (profiles/run_profile.sh: separate --pmc
    FETCH_SIZE / WRITE_SIZE passes of this same command; profiles/summarize.py)."""
    import glob
    # the committed summaries are of the default-precision run; an fp32-mode profile would be r*_pmc_summary_fp32.json
    # (the un-suffixed summaries of rounds 1-2 are of those rounds' default runs; from round 3 on every summary is suffixed)
    suffix = {"f16x2": "_f16x2", "bf16x3": "_bf16x3", "bf16x3_fast": "_bf16x3", "fp32": "_fp32"}[precision]
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary%s.json" % suffix)))
    if not files:
        return None, None
    d = json.load(open(files[-1]))
    tot = n = 0.0
    for k, v in d.items():
        if is_walk_kernel(k, precision):
            if "fetch_bytes_per_launch_x2corr" in v and "write_bytes_per_launch" in v:
                tot += (v["fetch_bytes_per_launch_x2corr"] + v["write_bytes_per_launch"]) * v["launches"]
                n += v["launches"]
    return (int(tot / n), "committed: " + os.path.relpath(files[-1], ROOT)) if n else (None, None)


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
