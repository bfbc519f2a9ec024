#!/usr/bin/env python3
"""Fine-tune loop stability probe (GPU box): 12 iterations of train_on_batch on one synthetic batch, printing losses, whether the
gradients are finite and their maximum, then every parameter whose gradient looks wrong.  Used to validate the opt-in hipGraph
replay of the decoder scans (LRP_TRAIN_GRAPH=1; OWNSTREAM=1 runs the loop on a non-default stream; DEC=adaptive|gridtd, B=, ITERS=, SYNC=1)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lrp_imagecaptioning_amd.explainers import CaptionModelSpec, ExplainImgCaptioningAdaptiveAttention, ExplainImgCaptioningGridTDModel
from lrp_imagecaptioning_amd.synthetic import adaptive_weights, gridtd_weights, images, vgg_weights
from lrp_imagecaptioning_amd.training import TrainingLRPInferenceAdaptive, TrainingLRPInferenceGridTD
B, T, V = int(os.environ.get("B", 32)), 21, 10000
kind = os.environ.get("DEC", "gridtd")
rs = np.random.RandomState(0)
w = vgg_weights(rs)
w.update(gridtd_weights(rs, 196, 512, 512, 512, V) if kind == "gridtd" else adaptive_weights(rs, 196, 512, 512, 512, V))
cls = ExplainImgCaptioningGridTDModel if kind == "gridtd" else ExplainImgCaptioningAdaptiveAttention
ex = cls(CaptionModelSpec(w), None, None, max_caption_length=T - 1, max_images=B)
tr = (TrainingLRPInferenceGridTD if kind == "gridtd" else TrainingLRPInferenceAdaptive)(ex, learning_rate=2e-4, drop_rate=0.5)
ex._engine.train_set_precision("bf16")
rs = np.random.RandomState(100)
X = torch.as_tensor(images(rs, B)).cuda()
cap_in = np.concatenate([np.full((B, 1), 1), rs.randint(2, V, size=(B, T - 1))], axis=1).astype(np.int32)
y = rs.randint(0, V, size=(B, T)).astype(np.int32)
import contextlib
ctx = torch.cuda.stream(torch.cuda.Stream()) if os.environ.get("OWNSTREAM") else contextlib.nullcontext()
with ctx:
  for it in range(int(os.environ.get('ITERS', 12))):
    l = tr.train_on_batch([cap_in, X], y)
    if os.environ.get("SYNC"): torch.cuda.synchronize()
    g = tr._grads
    print(it, [round(float(v), 3) for v in l], bool(torch.isfinite(g).all()), float(g.abs().max()), flush=True)
eng = ex._engine
g = tr._grads
for name, (off, n) in sorted(eng.train_layout.items(), key=lambda kv: kv[1][0]):
    seg = g[off:off + n]
    m = float(seg.abs().max())
    if m > 50 or not bool(torch.isfinite(seg).all()):
        print("BAD %-28s max %.4g" % (name, m))
