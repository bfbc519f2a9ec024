"""-m gpu: parity of the CNN half on TRAINED-LIKE weight statistics, through the engine, in every arithmetic mode.

Every other parity test draws He-normal kernels: dense Gaussians, thousands of comparable products per alpha1beta0 sum, so a
per-weight rounding error averages out.  The reference explains a trained VGG16 (models/model.py:420,
models/explainers.py:27-32): sparse, heavy-tailed kernels, sparse post-ReLU activations, concentrated relevance.  Here
the kernels keep 15 % / 5 % / 1 % of their entries with lognormal(sigma = 1 / 1.5 / 2.5) magnitudes, the biases are pushed
negative until >= 80 % of every channel's activations are zero (synthetic.vgg_weights_trained_like), and the relevance
entering the encoder is dense, one-hot, or the 20 largest features.

Bar (BASELINE.json): 1e-4 relative L1 on the raw (224, 224, 3) relevance against the float64 literal graph
(oracle/cnn_lrp_ref.py; RR:274-322 on TF float32 in the reference) — WHERE THE REFERENCE'S OWN ARITHMETIC IS THAT STABLE.
alpha1beta0 divides by Z+ = x.w+ + b, and with biases this negative some denominators pass close to zero: on the 1 % case
a float32 evaluation of the literal graph (what TensorFlow computes) is itself 7e-4 from the float64 one, and the decoder's
LRP amplifies a 2e-6 feature difference to 1e-4 ... 7e-4 on these features (measured, below).  The bound used is therefore
max(1e-4, 3 x the float32-vs-float64 distance of the reference graph on the same case) [measured on the 1 % case: fp32 mode
1.05 x, bf16x3 2.0 x that distance]: tied to the noise the reference has, never tighter than what its float32 arithmetic
can deliver.
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
    # the case is what it claims to be: >= 75 % of every conv's post-ReLU outputs are zero, kernels sparse
    for i in range(1, len(layers)):
        if layers[i - 1][0] == "conv":                        # inputs[i] = output of conv i - 1 (before any pool)
            assert float((inputs[i] > 0).double().mean()) < 0.25
    assert float((feat_ref > 0).mean()) < 0.25
    assert float((w["block3_conv2_W"] != 0).mean()) < density * 1.2
    R = _relevances(feat_ref)
    X3 = np.repeat(X, 3, 0)
    ref = C.analyze(layers, X3, R)
    assert np.isfinite(ref).all() and all(np.abs(r).sum() > 0 for r in ref)
    # the reference graph's own float32 noise on this case (TF float32 vs the float64 evaluation)
    ref32 = C.analyze(layers, X3, R, torch.float32)
    noise = dict(feat=rel_l1(C.forward(layers, X, torch.float32), feat_ref), dense=rel_l1(ref32[0], ref[0]),
                 onehot=rel_l1(ref32[1], ref[1]), top20=rel_l1(ref32[2], ref[2]))
    report("stress_%s_reference_fp32_noise" % name, **noise)
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
        assert r["feat"] < max(1e-5, 3 * noise["feat"]), (prec, r, noise)
        for k in ("dense", "onehot", "top20"):
            assert r[k] < max(TOL, 3 * noise[k]), (prec, k, r, noise)
    r = res["f16x2"]
    assert max(r["dense"], r["onehot"], r["top20"]) < max(F16X2_WORST_CASE, 3 * noise["dense"]), r


def test_trained_like_weights_through_the_decoder():
    """The same statistics end to end: decoder LRP (AdaptiveOracle, pinned by the reference's own outputs) -> CNN LRP, one
    image, three words, every mode.  On these sparse features the decoder's LRP is ill-conditioned — it divides by cell
    states and pre-activations, and a 2e-6 relative difference in the CNN features moves the heat-map by 1e-4 ... 7e-4
    [measured on CPU: float32 vs float64 encoder forward in front of the same oracle decoder; the amplification varies by
    two orders of magnitude with the rounding pattern] — so 'features equal to float64 within 1e-5' and 'explanation equal
    given the features' are checked separately, and the end-to-end figure is recorded with the amplification beside it:
      * features vs the float64 forward: 1e-5;
      * the oracle decoder run on the ENGINE's features -> float64 literal graph vs the engine's heat-map: 1e-4
        (decoder kernels + CNN walk on decoder-produced, concentrated, signed relevance);
      * the CNN half alone on the engine's own R_feat: 1e-4."""
    from lrp_imagecaptioning_amd.engine import LRPEngine
    from oracle.decoder_ref import AdaptiveOracle
    w, X = _weights(0.05, 1.5)
    V = 500
    w.update(adaptive_weights(np.random.RandomState(5), 196, 512, 512, 512, V))
    cap = captions(np.random.RandomState(6), 1, 3, V)
    layers = C.vgg_layers(w, VGG16_CFG)
    toks = (1, 3)
    feat_ref = C.forward(layers, X)
    dec = AdaptiveOracle(w, 196, 512, 512, 512)
    dec.forward(feat_ref.astype(np.float32), cap[0])
    ref64 = {t: C.analyze(layers, X, dec.explain(t)[0])[0] for t in toks}
    eng = LRPEngine(decoder="adaptive", V=V, max_images=1, max_tokens=2, max_caption_len=4)
    eng.set_weights(w)
    for prec in ("bf16x3", "fp32", "f16x2"):
        eng.set_precision(prec)
        eng.encode_images(X)
        feat = eng.get_features().cpu().numpy().reshape(feat_ref.shape)
        e_feat = rel_l1(feat, feat_ref)
        eng.decoder_forward(cap)
        out, Rf, _, _ = eng.explain_tokens([0, 0], list(toks), want_R_feat=True)
        out, Rf = out.cpu().numpy(), Rf.cpu().numpy().reshape(2, 14, 14, 512)
        cnn_only = C.analyze(layers, np.repeat(X, 2, 0), Rf)
        e_cnn = max(rel_l1(out[i], cnn_only[i]) for i in range(2))
        dec_e = AdaptiveOracle(w, 196, 512, 512, 512)                 # the oracle decoder on the engine's features
        dec_e.forward(feat.astype(np.float32), cap[0])
        e_given = max(rel_l1(out[i], C.analyze(layers, X, dec_e.explain(t)[0])[0]) for i, t in enumerate(toks))
        e_all = max(rel_l1(out[i], ref64[t]) for i, t in enumerate(toks))
        report("stress_decoder_" + prec, features=e_feat, cnn_half_on_engine_R_feat=e_cnn, given_engine_features=e_given,
               end_to_end_vs_float64_pipeline=e_all, amplification=e_all / max(e_feat, 1e-30))
        assert e_feat < 1e-5, (prec, e_feat)
        if prec != "f16x2":
            assert e_cnn < TOL, (prec, e_cnn)
            assert e_given < TOL, (prec, e_given)
        else:
            assert e_cnn < F16X2_WORST_CASE, e_cnn


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
