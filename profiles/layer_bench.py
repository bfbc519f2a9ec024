#!/usr/bin/env python3
"""Per-launch timing of the conv-LRP reverse walk (HIP events around each launch, on the
launch stream) for the BASELINE workload: prints ms and algorithmic TFLOP/s per VGG16 layer.
Usage (GPU box): python profiles/layer_bench.py [--batch 32] [--tokens 10] [--reps 3]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--tokens", type=int, default=10)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--vocab", type=int, default=10000)
    ap.add_argument("--precision", default="bf16x3", choices=["fp32", "bf16x3", "f16x2"])
    a = ap.parse_args()
    import torch
    from bench import synth_weights
    from lrp_imagecaptioning_amd.engine import LRPEngine
    from lrp_imagecaptioning_amd.synthetic import VGG16_CFG, captions, images
    B, T, V = a.batch, a.tokens, a.vocab
    eng = LRPEngine(decoder="adaptive", V=V, max_images=B, max_tokens=B * T, max_caption_len=T + 1)
    eng.set_weights(synth_weights(0, V))
    eng.set_precision(a.precision)
    rs = np.random.RandomState(1)
    X = torch.as_tensor(images(rs, B)).cuda()
    caps = captions(rs, B, T, V)
    img_idx = [b for b in range(B) for _ in range(T)]
    tpos = [t for _ in range(B) for t in range(1, T + 1)]
    out = torch.empty((B * T, 224, 224, 3), dtype=torch.float32, device="cuda")

    def step():
        eng.encode_images(X)
        eng.decoder_forward(caps)
        eng.explain_tokens(img_idx, tpos, out=out)

    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        step()
    torch.cuda.synchronize()
    ms_step = (time.perf_counter() - t0) / a.reps * 1e3
    # phase split
    def timed(fn):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e3
    t_enc = timed(lambda: eng.encode_images(X))
    t_dec = timed(lambda: eng.decoder_forward(caps))
    t_dex = timed(lambda: eng.decoder_explain(img_idx, tpos, want_attention=False, want_r_words=False))
    eng.profile_enable(True)
    acc = None
    for _ in range(a.reps):
        eng.explain_tokens(img_idx, tpos, out=out)
        torch.cuda.synchronize()
        rec = eng.profile_records()
        acc = rec if acc is None else [(m0 + m1, f) for (m0, f), (m1, _) in zip(acc, rec)]
    eng.profile_enable(False)
    names = [c[0] for c in VGG16_CFG][::-1]
    tot_ms = tot_fl = 0.0
    print("ms/step %.2f  (encode %.2f, decoder fwd %.2f, decoder explain %.2f)" % (ms_step, t_enc, t_dec, t_dex))
    for nm, (ms, fl) in zip(names, acc):
        ms /= a.reps
        tot_ms += ms
        tot_fl += fl
        print("  bwd %-13s %8.3f ms  %7.1f GFLOP  %6.1f TF/s" % (nm, ms, fl / 1e9, fl / ms / 1e9))
    print("  conv-LRP total  %8.3f ms  %7.1f GFLOP  %6.1f TF/s  (%.1f%% of 157.3)" % (
        tot_ms, tot_fl / 1e9, tot_fl / tot_ms / 1e9, 100 * tot_fl / tot_ms / 1e9 / 157.3))


if __name__ == "__main__":
    main()
