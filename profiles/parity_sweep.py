#!/usr/bin/env python3
"""Heat-map parity of the CNN half over many seeds, per arithmetic mode (GPU box): VGG16 224x224, one image and two
(--tokens) relevance maps per seed against the float64 literal graph (oracle/cnn_lrp_ref.py).  Prints one line per seed and the
worst / median per mode.  Usage: python profiles/parity_sweep.py [--seeds 13] [--modes bf16x3,f16x2,fp32]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=13)
    ap.add_argument("--modes", default="bf16x3,f16x2,fp32")
    ap.add_argument("--tokens", type=int, default=2, help="relevance maps per seed; >= 12 puts every layer on its large-tile path "
                                                         "(compact pool interfaces, 256 x 256 tiles from ~100 on)")
    a = ap.parse_args()
    from lrp_imagecaptioning_amd.engine import LRPEngine
    from lrp_imagecaptioning_amd.synthetic import VGG16_CFG, images, vgg_weights
    from oracle import cnn_lrp_ref as C
    modes = a.modes.split(",")
    T = a.tokens
    eng = LRPEngine(decoder="adaptive", V=50, H=32, E=32, max_images=1, max_tokens=T, max_caption_len=4)
    res = {m: [] for m in modes}
    for seed in range(a.seeds):
        rs = np.random.RandomState(seed)
        w = vgg_weights(rs)
        layers = C.vgg_layers(w, VGG16_CFG)
        X = images(rs, 1)
        feat = C.forward(layers, X)
        R = (rs.standard_normal((T, 14, 14, 512)) * feat).astype(np.float32)
        ref = C.analyze(layers, np.repeat(X, T, axis=0), R)
        eng.set_weights({k: v for k, v in w.items()})
        line = []
        for m in modes:
            eng.set_precision(m)
            eng.encode_images(X)
            out = eng.cnn_explain([0] * T, R).cpu().numpy()
            e = max(float(np.abs(out[i] - ref[i]).sum() / np.abs(ref[i]).sum()) for i in range(T))
            res[m].append(e)
            line.append("%s %.2e" % (m, e))
        print("seed %2d  %s" % (seed, "   ".join(line)), flush=True)
    for m in modes:
        v = np.array(res[m])
        print("%-12s worst %.2e  median %.2e  best %.2e  (%d seeds)" % (m, v.max(), np.median(v), v.min(), len(v)))


if __name__ == "__main__":
    main()
