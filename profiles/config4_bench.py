#!/usr/bin/env python3
"""BASELINE config 4 on ONE GPU's share: grid-TD decoder + ResNet-101 encoder, LRP-alpha1beta0, 32 images x 10 words
(the config quotes batch=128 on 4 GPUs = 32 per GPU).  Prints ms/step and heat-maps/s; not the headline bench."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from lrp_imagecaptioning_amd.engine import LRPEngine
    from lrp_imagecaptioning_amd.synthetic import RESNET101_STACKS, captions, gridtd_weights, images, resnet_weights
    B, T, V = int(os.environ.get("B", 32)), 10, 10000
    rs = np.random.RandomState(0)
    w = resnet_weights(rs)
    w.update(gridtd_weights(rs, 49, 2048, 512, 512, V))
    eng = LRPEngine(decoder="gridtd", img_hw=(224, 224), L=49, D=2048, H=512, E=512, V=V, max_images=B, max_tokens=B * T,
                    max_caption_len=T + 1, resnet={"stem": 64, "stacks": RESNET101_STACKS})
    eng.set_weights(w)
    X = torch.as_tensor(images(rs, B)).cuda()
    caps = captions(rs, B, T, V)
    idx = [b for b in range(B) for _ in range(T)]
    tt = [t for _ in range(B) for t in range(1, T + 1)]
    out = torch.empty((B * T, 224, 224, 3), dtype=torch.float32, device="cuda")

    def step():
        eng.encode_images(X)
        eng.decoder_forward(caps)
        eng.explain_tokens(idx, tt, out=out)

    def timed(fn):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e3
    step()
    n = 3
    ms = timed(lambda: [step() for _ in range(n)]) / n
    handles = int(os.environ.get("HANDLES", 2))
    if handles > 1:                                     # two batches in flight (pipeline.py), like bench.py's default
        from lrp_imagecaptioning_amd.pipeline import LRPPipeline
        pipe = LRPPipeline(handles, decoder="gridtd", img_hw=(224, 224), L=49, D=2048, H=512, E=512, V=V, max_images=B,
                           max_tokens=B * T, max_caption_len=T + 1, resnet={"stem": 64, "stacks": RESNET101_STACKS})
        pipe.set_weights(w)
        outs = [torch.empty_like(out) for _ in range(handles)]
        for _ in range(handles):
            pipe.explain_batch(X, caps, idx, tt, out=outs[pipe.next_slot])
        pipe.synchronize()
        n2 = 6
        ms2 = timed(lambda: [[pipe.explain_batch(X, caps, idx, tt, out=outs[pipe.next_slot]) for _ in range(n2)], pipe.synchronize()]) / n2
        print("config4 with %d handles in flight: %.2f ms/step = %.1f heat-maps/s" % (handles, ms2, B * T / ms2 * 1e3))
    eng.decoder_explain(idx, tt, want_attention=False, want_r_words=False)     # (first call of this entry: its output tensor is allocated)
    print("config4 (grid-TD + ResNet-101, B=%d, T=%d): %.2f ms/step = %.1f heat-maps/s; encode %.2f, decoder fwd %.2f, "
          "decoder explain %.2f ms; workspace %.1f GB" % (
              B, T, ms, B * T / ms * 1e3, timed(lambda: eng.encode_images(X)), timed(lambda: eng.decoder_forward(caps)),
              timed(lambda: eng.decoder_explain(idx, tt, want_attention=False, want_r_words=False)),
              eng.workspace_bytes / 1e9))
    assert torch.isfinite(out).all()


if __name__ == "__main__":
    main()
