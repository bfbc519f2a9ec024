#!/usr/bin/env python3
"""Single-image latency loop of bench.py's `latency` block, alone (for rocprofv3 --kernel-trace): B = 1, T = 10."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_weights
from lrp_imagecaptioning_amd.engine import LRPEngine
from lrp_imagecaptioning_amd.synthetic import captions, images
T, V = 10, 10000
eng = LRPEngine(decoder="adaptive", V=V, max_images=1, max_tokens=T, max_caption_len=T + 1)
eng.set_weights(synth_weights(0, V))
rs = np.random.RandomState(1)
X = torch.as_tensor(images(rs, 1)).cuda()
caps = captions(rs, 1, T, V)
out = torch.empty((T, 224, 224, 3), dtype=torch.float32, device="cuda")
def phase(fn):
    torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e3
for r in range(5):
    te = phase(lambda: eng.encode_images(X)); td = phase(lambda: eng.decoder_forward(caps))
    tx = phase(lambda: eng.explain_tokens([0] * T, list(range(1, T + 1)), out=out))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.encode_images(X); eng.decoder_forward(caps); eng.explain_tokens([0] * T, list(range(1, T + 1)), out=out)
    torch.cuda.synchronize(); tt = (time.perf_counter() - t0) * 1e3
    print("rep %d: encode %.3f  decoder_forward %.3f  explain_tokens %.3f  (phases synchronised) | one shot %.3f ms" % (r, te, td, tx, tt))
