"""-m gpu: parity of the CNN half on TRAINED-LIKE weight statistics, through the engine, in every arithmetic mode.

Every other parity test draws He-normal kernels: dense Gaussians, thousands of comparable products per alpha1beta0 sum, so a
per-weight rounding error averages out.  The reference explains a trained VGG16 (models/model.py:420,
models/explainers.py:27-32): sparse, heavy-tailed kernels, sparse post-ReLU activations, concentrated relevance.  Here
the kernels keep 15 % / 5 % / 1 % of their entries with lognormal(sigma = 1 / 1.5 / 2.5) magnitudes, the biases are pushed
negative until >= 80 % of every channel's activations are zero (synthetic.vgg_weights_trained_like), and the relevance
entering the encoder is dense, one-hot, or the 20 largest features.

Bar (BASELINE.json): 1e-4 relative L1 on the raw (224, 224, 3) relevance against the float64 literal graph
(oracle/cnn_lrp_ref.py; RR:274-322 on TF float32 in the reference).
  * bf16x3 (the library default) and fp32 must hold it on every case: 16 / 24 mantissa bits on BOTH operands of every
    product, a worst case that does not depend on the weights.
  * f16x2 (opt-in fast mode: ONE fp16 per weight below the top block) is MEASURED and recorded, and only bounded by its own
    worst case (2^-12 per product and layer; bound used 2e-3): it is expected to leave the 1e-4 bar on the sparse cases,
    which is why it is not the default (include/lrp_hip.h, lrp_set_precision).
"""
import numpy as np
import pytest
import torch

from conftest import rel_l1
from gpu_util import report
from lrp_imagecaptioning_amd.synthetic import VGG16_CFG, adaptive_weights, captions, images, vgg_weights_trained_like
from oracle import cnn_lrp_ref as C

pytestmark = pytest.mark.gpu
TOL = 1e-4
F16X2_WORST_CASE = 2e-3          # ~ 9 two-term layers x 2^-12, nothing averaging: a sanity bound, not a parity claim

CASES = [("d15_s1.0", 0.15, 1.0), ("d05_s1.5", 0.05, 1.5), ("d01_s2.5", 0.01, 2.5)]


def _weights(density, sigma, seed=1):
    X = images(np.random.RandomState(0), 1)
    w = vgg_weights_trained_like(np.random.RandomState(seed), VGG16_CFG, density, sigma, 0.2, X)
    return w, X


def _relevances(feat):
    """dense N(0,1) * feat | one-hot at the largest feature | the 20 largest features (their own values)"""
    rs = np.random.RandomState(2)
    dense = (rs.standard_normal(feat.shape) * feat).astype(np.float32)
    flat = feat.reshape(-1)
    order = np.argsort(flat)[::-1]
    onehot = np.zeros_like(flat)
    onehot[order[0]] = flat[order[0]]
    top20 = np.zeros_like(flat)
    top20[order[:20]] = flat[order[:20]]
    return np.concatenate([dense, onehot.reshape(feat.shape), top20.reshape(feat.shape)]).astype(np.float32)


@pytest.mark.parametrize("name,density,sigma", CASES)
def test_vgg16_trained_like_weights_every_mode(name, density, sigma):
    from lrp_imagecaptioning_amd.engine import LRPEngine
    w, X = _weights(density, sigma)
    layers = C.vgg_layers(w, VGG16_CFG)
    feat_ref, inputs = C.forward(layers, X, return_inputs=True)
    # the case is what it claims to be: >= 75 % of the post-ReLU activations feeding every conv are zero, kernels sparse
    for L, x in zip(layers[1:], inputs[1:]):
        if L[0] == "conv":
            assert float((x > 0).double().mean()) < 0.25
    assert float((w["block3_conv2_W"] != 0).mean()) < density * 1.2
    R = _relevances(feat_ref)
    ref = C.analyze(layers, np.repeat(X, 3, 0), R)
    assert np.isfinite(ref).all() and all(np.abs(r).sum() > 0 for r in ref)
    eng = LRPEngine(decoder="adaptive", cnn_cfg=VGG16_CFG, img_hw=(224, 224), L=196, D=512, H=32, E=32, V=50,
                    max_images=1, max_tokens=3, max_caption_len=4)
    eng.set_weights(w)
    res = {}
    for prec in ("bf16x3", "fp32", "f16x2"):
        eng.set_precision(prec)
        eng.encode_images(X)                                  # (a mode change drops the caches)
        feat = eng.get_features().cpu().numpy().reshape(feat_ref.shape)
        out = eng.cnn_explain([0, 0, 0], R).cpu().numpy()
        assert np.isfinite(out).all()
        res[prec] = dict(feat=rel_l1(feat, feat_ref), dense=rel_l1(out[0], ref[0]), onehot=rel_l1(out[1], ref[1]),
                         top20=rel_l1(out[2], ref[2]))
        report("stress_%s_%s" % (name, prec), **res[prec])
    for prec in ("bf16x3", "fp32"):
        r = res[prec]
        assert r["feat"] < 1e-5, (prec, r)
        assert max(r["dense"], r["onehot"], r["top20"]) < TOL, (prec, r)
    r = res["f16x2"]
    assert max(r["dense"], r["onehot"], r["top20"]) < F16X2_WORST_CASE, r


def test_trained_like_weights_through_the_decoder():
    """The same statistics end to end: decoder LRP (AdaptiveOracle, pinned by the reference's own outputs) -> CNN LRP, one
    image, three words, default arithmetic and fp32, against oracle decoder + float64 literal graph."""
    from lrp_imagecaptioning_amd.engine import LRPEngine
    from oracle.decoder_ref import AdaptiveOracle
    w, X = _weights(0.05, 1.5)
    V = 500
    w.update(adaptive_weights(np.random.RandomState(5), 196, 512, 512, 512, V))
    cap = captions(np.random.RandomState(6), 1, 3, V)
    layers = C.vgg_layers(w, VGG16_CFG)
    feat = C.forward(layers, X).astype(np.float32)
    dec = AdaptiveOracle(w, 196, 512, 512, 512)
    dec.forward(feat, cap[0])
    ref = {t: C.analyze(layers, X, dec.explain(t)[0])[0] for t in (1, 3)}
    eng = LRPEngine(decoder="adaptive", V=V, max_images=1, max_tokens=2, max_caption_len=4)
    eng.set_weights(w)
    for prec in ("bf16x3", "fp32", "f16x2"):
        eng.set_precision(prec)
        eng.encode_images(X)
        eng.decoder_forward(cap)
        out = eng.explain_tokens([0, 0], [1, 3])[0].cpu().numpy()
        errs = [rel_l1(out[0], ref[1]), rel_l1(out[1], ref[3])]
        report("stress_decoder_" + prec, max_rel_l1=max(errs))
        assert max(errs) < (TOL if prec != "f16x2" else F16X2_WORST_CASE), (prec, errs)


def test_precision_change_drops_the_encode_caches():
    """lrp_set_precision: the gates belong to the arithmetic they were computed in; after a mode CHANGE an explain call
    without a new encode must be refused (LRP_ERR_STATE), the same mode again must not drop anything."""
    from lrp_imagecaptioning_amd.engine import LRPEngine
    cfg = [("c1", 3, 8, True), ("c2", 8, 16, False)]
    rs = np.random.RandomState(3)
    from lrp_imagecaptioning_amd.synthetic import vgg_weights
    w = vgg_weights(rs, cfg)
    eng = LRPEngine(decoder="adaptive", cnn_cfg=cfg, img_hw=(16, 16), L=64, D=16, H=32, E=32, V=50, max_images=1, max_tokens=1,
                    max_caption_len=4)
    eng.set_weights(w)
    X = rs.uniform(-100, 100, size=(1, 16, 16, 3)).astype(np.float32)
    R = rs.standard_normal((1, 64, 16)).astype(np.float32)
    assert eng.precision == "bf16x3"                         # the library default
    eng.encode_images(X)
    a = eng.cnn_explain([0], R).clone()
    eng.set_precision("bf16x3")                              # no change: caches stay
    assert torch.equal(eng.cnn_explain([0], R), a)
    eng.set_precision("fp32")
    with pytest.raises(RuntimeError):
        eng.cnn_explain([0], R)
    eng.encode_images(X)
    b = eng.cnn_explain([0], R)
    assert rel_l1(b.cpu().numpy(), a.cpu().numpy()) < 1e-4
