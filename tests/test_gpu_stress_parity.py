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


DECODER_SEEDS = [(1, 5, 6), (2, 7, 8), (3, 9, 10), (4, 11, 12)]      # (encoder kernels, decoder weights, caption)


def _undecidable_units(dec, w, feat_tol=1e-5):
    """Units (location i, channel h) of relu(F.W_if + b_if) (E:378-381) whose float64 pre-activation is smaller than
    `feat_tol` x sum_d |F[i,d] W[d,h]| + |b_h|: an evaluation whose features are only `feat_tol` close to float64 — the
    reference's own float32 forward is 2e-6 away — cannot decide that ReLU."""
    mag = np.abs(dec.F) @ np.abs(w["image_features_W"]) + np.abs(w["image_features_b"])
    return {tuple(u) for u in np.argwhere(np.abs(dec.if_pre) < feat_tol * mag)}


def _units(us):
    return sorted([int(i), int(h)] for i, h in us)


def _flips(dec_a, dec_b):
    return {tuple(u) for u in np.argwhere((dec_a.if_pre > 0) != (dec_b.if_pre > 0))}


@pytest.mark.parametrize("seeds", DECODER_SEEDS, ids=lambda s: "seed%d" % s[0])
def test_trained_like_weights_through_the_decoder(seeds):
    """The same statistics end to end: decoder LRP (AdaptiveOracle, pinned by the reference's own outputs) -> CNN LRP, one
    image, every default-grade mode, four draws of encoder kernels / decoder weights / caption.

    On these sparse features the reference's algorithm has ONE discrete switch inside float32 noise, located in round 4
    (DESIGN section 6): the ReLU of `_image_features = relu(F.W_if + b_if)` (E:378-381).  The attention-sum rule gives a
    unit (i, h) the relevance r_ctx[h].alpha_i.V[i,h]/st(ctx[h]) (E:648-653) and the image_features rule divides it by
    st(pre-activation) again (E:654-659): the product is r_ctx[h].alpha_i/ctx[h] x V/(V + 1e-7) — a FULL-SIZE share however
    small V[i,h] is, and exactly 0 when the pre-activation is <= 0.  A pre-activation of +9e-7 in the reference's float32
    forward and -3e-7 in float64 (seed 1: location 19, channel 295) therefore moves the heat-map by 1e-4 ... 7e-4 — the
    bimodal figure VERDICT r3 reproduced — while every other decoder quantity agrees to 4e-7.

    Asserted, bf16x3 (default) and fp32:
      * features vs the float64 forward < 1e-5;
      * the engine decides differently from float64 only on units no evaluation at that feature tolerance can decide
        (_undecidable_units), and with those decisions aligned in the float64 reference (like an exact max-pool tie)
        the END-TO-END heat-map is within 1e-4;
      * un-aligned, END TO END: < max(1e-4, 3 x the reference's own float32 noise) where that noise is the distance of
        the float32 pipeline (float32 C.forward -> AdaptiveOracle -> float32 literal graph: what TF/numpy compute) from
        the float64 one, taken over both states of the switch when the engine and the float32 pipeline fell on
        different sides of it;
      * the oracle decoder on the ENGINE's features -> float64 literal graph vs the engine's heat-map < 1e-4;
      * (first seed) the CNN half alone on the engine's own R_feat < 1e-4; f16x2 recorded, bounded by its worst case."""
    from lrp_imagecaptioning_amd.engine import LRPEngine
    from oracle.decoder_ref import AdaptiveOracle
    ws, ds, cs = seeds
    first = seeds == DECODER_SEEDS[0]
    w, X = _weights(0.05, 1.5, ws)
    V = 500
    w.update(adaptive_weights(np.random.RandomState(ds), 196, 512, 512, 512, V))
    cap = captions(np.random.RandomState(cs), 1, 3, V)
    layers = C.vgg_layers(w, VGG16_CFG)
    toks = (1, 3) if first else (3,)
    nt = len(toks)
    Xn = np.repeat(X, nt, 0)

    def decoder_on(feat):
        d = AdaptiveOracle(w, 196, 512, 512, 512)
        d.forward(feat.astype(np.float32), cap[0])
        return d

    def heatmaps(dec, dtype=torch.float64):
        return C.analyze(layers, Xn, np.concatenate([dec.explain(t)[0] for t in toks]), dtype)

    def dist(a, b):
        return max(rel_l1(a[i], b[i]) for i in range(nt))

    feat_ref = C.forward(layers, X)
    dec64 = decoder_on(feat_ref)
    ref64 = heatmaps(dec64)
    undecidable = _undecidable_units(dec64, w)
    # the reference's own arithmetic: float32 forward -> numpy decoder -> float32 literal graph
    dec32 = decoder_on(C.forward(layers, X, torch.float32))
    noise32 = dist(heatmaps(dec32, torch.float32), ref64)
    flips32 = _flips(dec32, dec64)
    assert flips32 <= undecidable, (flips32, undecidable)

    def aligned_reference(flips, dec_side):
        """float64 pipeline with the listed units' ReLU decisions taken from `dec_side`"""
        d = decoder_on(feat_ref)
        for u in flips:
            d.if_pre[u], d.Vfeat[u] = dec_side.if_pre[u], dec_side.Vfeat[u]
        return heatmaps(d)

    report("stress_e2e_reference", seed=ws, tokens=list(toks), float32_pipeline=noise32,
           float32_pipeline_flipped_units=_units(flips32), undecidable_units=_units(undecidable),
           min_margin=float(np.min(np.abs(dec64.if_pre) / (np.abs(dec64.F) @ np.abs(w["image_features_W"])
                                                            + np.abs(w["image_features_b"])))))
    eng = LRPEngine(decoder="adaptive", V=V, max_images=1, max_tokens=2, max_caption_len=4)
    eng.set_weights(w)
    for prec in (("bf16x3", "fp32", "f16x2") if first else ("bf16x3", "fp32")):
        eng.set_precision(prec)
        eng.encode_images(X)
        feat = eng.get_features().cpu().numpy().reshape(feat_ref.shape)
        e_feat = rel_l1(feat, feat_ref)
        eng.decoder_forward(cap)
        out, Rf, _, _ = eng.explain_tokens([0] * nt, list(toks), want_R_feat=True)
        out, Rf = out.cpu().numpy(), Rf.cpu().numpy().reshape(nt, 14, 14, 512)
        dec_e = decoder_on(feat)                                      # the oracle decoder on the engine's features
        e_given = dist(out, heatmaps(dec_e))
        e_all = dist(out, ref64)
        flips = _flips(dec_e, dec64)
        aligned = aligned_reference(flips, dec_e) if flips else ref64
        e_aligned = dist(out, aligned)
        # both states of the reference's switch: the float32 pipeline's side and, where the engine fell on the other
        # side of an undecidable unit, the float64 reference moved to that side
        noise = max(noise32, dist(aligned, ref64))
        rec = dict(seed=ws, features=e_feat, given_engine_features=e_given, end_to_end_vs_float64_pipeline=e_all,
                   end_to_end_decisions_aligned=e_aligned, flipped_units=_units(flips),
                   reference_float32_noise=noise32, reference_noise_both_states=noise)
        if first:
            cnn_only = C.analyze(layers, Xn, Rf)
            rec["cnn_half_on_engine_R_feat"] = e_cnn = dist(out, cnn_only)
        report("stress_decoder_" + prec, **rec)
        assert e_feat < 1e-5, (prec, e_feat)
        if prec != "f16x2":
            assert flips <= undecidable, (prec, flips, undecidable)
            assert e_given < TOL, (prec, e_given)
            assert e_aligned < TOL, (prec, rec)
            assert e_all < max(TOL, 3 * noise), (prec, rec)
            if first:
                assert e_cnn < TOL, (prec, e_cnn)
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
