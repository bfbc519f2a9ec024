"""-m gpu: the gradient baselines through the C ABI.
Decoder half: lrp_decoder_gradient against goldens produced by the reference's own `_lstm_decoder_backward`
(E:780-832, E:1452-1532; tests/golden/make_golden.py --only grad).  CNN half: lrp_cnn_walk against the float64
per-layer autograd restatement of iNNvestigate's Gradient / InputTimesGradient / GuidedBackprop
(oracle/cnn_lrp_ref.gradient_analyze)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_l1
from gpu_util import report
from lrp_imagecaptioning_amd.synthetic import VGG16_CFG, decoder_case, vgg_weights
from oracle import cnn_lrp_ref as C

pytestmark = pytest.mark.gpu
TOL = 1e-4

GRAD_GOLDENS = ["adaptive_grad_small_s0", "adaptive_grad_small_s1", "adaptive_grad_small_s2", "gridtd_grad_small_s0",
                "gridtd_grad_small_s1", "gridtd_grad_small_s2", "adaptive_grad_full_s0", "gridtd_grad_full_s0"]


@pytest.mark.parametrize("name", GRAD_GOLDENS)
def test_decoder_gradient_matches_reference(name):
    from lrp_imagecaptioning_amd.engine import LRPEngine
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    kind = str(g["kind"])
    L, D, H, E, V, T = [int(x) for x in g["dims"]]
    if "feat" in g.files:
        w = {k[2:]: g[k] for k in g.files if k.startswith("w_")}
        feat, cap = g["feat"], [int(c) for c in g["caption"]]
    else:
        w, feat, cap = decoder_case(kind, int(g["seed"]), L, D, H, V, T)
    toks = [int(t) for t in g["tokens"]]
    # two copies of the caption as a batch of 2 (the second image's units must not disturb the first's)
    gsz = int(round(np.sqrt(L)))           # the CNN is not exercised: a 1-conv stub encoder whose output matches (L, D)
    eng = LRPEngine(decoder=kind, cnn_cfg=[("c1", 3, D, False)], img_hw=(gsz, gsz), L=L, D=D, H=H, E=E, V=V, max_images=2,
                    max_tokens=2 * len(toks), max_caption_len=len(cap) + 1)
    eng.set_weights({k: v for k, v in w.items()})
    eng.set_features(np.stack([feat.reshape(L, D)] * 2))
    eng.decoder_forward([cap, cap])
    d, rw = eng.decoder_gradient([0] * len(toks) + [1] * len(toks), toks + toks)
    d, rw = d.cpu().numpy(), rw.cpu().numpy()
    worst = 0.0
    for j, t in enumerate(toks):
        ref = g["d_feat"][j].reshape(L, D)
        e = max(rel_l1(d[j], ref), rel_l1(d[len(toks) + j], ref))
        worst = max(worst, e)
        r = g["r_words_t%d" % t]
        assert np.abs(rw[j, :t] - r).sum() <= 1e-3 * np.abs(r).sum() + 1e-7
        assert (rw[j, t:] == 0).all()
    report("decoder_gradient", case=name, rel_l1=worst)
    assert worst < TOL, worst


CFG = [("c1", 3, 16, False), ("c2", 16, 16, True), ("c3", 16, 32, False), ("c4", 32, 32, True), ("c5", 32, 64, False)]


@pytest.mark.parametrize("walk", ["gradient", "input_x_gradient", "guided_backprop"])
@pytest.mark.parametrize("cfg,hw,nb", [(CFG, 16, 3), (VGG16_CFG, 224, 1)])
def test_cnn_gradient_walks(walk, cfg, hw, nb):
    from lrp_imagecaptioning_amd.engine import LRPEngine
    rs = np.random.RandomState(len(cfg) + hw)
    w = vgg_weights(rs, cfg, bias_std=0.05)
    layers = C.vgg_layers(w, cfg)
    X = rs.uniform(-120, 130, size=(nb, hw, hw, 3)).astype(np.float32)
    feat = C.forward(layers, X)
    h, ww, c = feat.shape[1:]
    eng = LRPEngine(decoder="adaptive", cnn_cfg=cfg, img_hw=(hw, hw), L=h * ww, D=c, H=8, E=8, V=8, max_images=nb,
                    max_tokens=2 * nb, max_caption_len=2)
    eng.set_weights(w)
    idx = list(range(nb)) + list(range(nb))
    head = rs.standard_normal((2 * nb, h, ww, c)).astype(np.float32)
    ref = C.gradient_analyze(layers, X[idx], head, walk)
    # A gradient through ReLUs and max-pools is only piecewise continuous in the forward: where a pre-activation is zero
    # to the forward's own accuracy the ReLU decision can differ from float64's, and ONE such flip switches a whole
    # gradient path (plain torch float32 on the CPU is 1e-3 ... 4e-3 from float64 at VGG16 size for the same reason; LRP
    # does not care — a unit with a ~ 0 carries ~ 0 relevance).  The walk's arithmetic is what is under test:
    #   * tight bar with the forward that flips least (exact fp32 MFMA);
    #   * the default forward (fp16 pairs: the same 7e-7 on the features, but ~4x the absolute error on the near-zero
    #     pre-activations that decide flips) must stay within what a few flips cost, 1e-2.
    for prec, bar in (("fp32", TOL), ("bf16x3", TOL if hw < 100 else 1e-2)):
        eng.set_precision(prec)
        eng.encode_images(X)
        out = eng.cnn_walk(idx, head, walk).cpu().numpy()
        errs = [rel_l1(out[i], ref[i]) for i in range(2 * nb)]
        report("cnn_walk", walk=walk, prec=prec, case=[len(cfg), hw, nb], rel_l1=max(errs))
        assert max(errs) < bar, (prec, errs)
    # the LRP walk through the same entry point is lrp_cnn_explain
    if walk == "gradient" and hw == 16:
        R = (head * feat[idx]).astype(np.float32)
        assert rel_l1(eng.cnn_walk(idx, R, "lrp").cpu().numpy(), eng.cnn_explain(idx, R).cpu().numpy()) == 0.0


def _small_caption_setup(kind, rs):
    from lrp_imagecaptioning_amd.explainers import CaptionModelSpec
    from lrp_imagecaptioning_amd.synthetic import adaptive_weights, gridtd_weights
    H, V, hw = 32, 50, 16
    w = vgg_weights(rs, CFG, bias_std=0.05)
    L, D = 16, 64
    w.update((adaptive_weights if kind == "adaptive" else gridtd_weights)(rs, L, D, H, H, V))
    spec = CaptionModelSpec(w, img_encoder="vgg16", hidden_dim=H, embedding_dim=H, L=L, D=D, vocab_size=V, cnn_cfg=CFG,
                            img_hw=(hw, hw))
    X = rs.uniform(-120, 130, size=(1, hw, hw, 3)).astype(np.float32)
    return w, spec, X, (L, D, H)


@pytest.mark.parametrize("kind", ["adaptive", "gridtd"])
def test_reference_gradient_classes(kind):
    """The six baseline engines of the reference (E:667-993, E:1322-1700) end to end on a small model, each against
    decoder-gradient oracle -> CNN-walk oracle (and the Grad-CAM gate for the guided variant)."""
    import lrp_imagecaptioning_amd.explainers as EX
    from lrp_imagecaptioning_amd.postprocess import grad_cam
    from oracle.decoder_grad_ref import AdaptiveGradOracle, GridTDGradOracle
    rs = np.random.RandomState(5)
    w, spec, X, (L, D, H) = _small_caption_setup(kind, rs)
    cap = [7, 19, 3, 1]
    layers = C.vgg_layers(w, CFG)
    feat = C.forward(layers, X).astype(np.float32)
    o = (AdaptiveGradOracle if kind == "adaptive" else GridTDGradOracle)(w, L, D, H, H)
    o.forward(feat, cap)
    names = {"adaptive": ("ExplainImgCaptioningAdaptiveAttentionGradient", "ExplainImgCaptioningAdaptiveAttentionInputTimesGradient",
                          "ExplainImgCaptioningAdaptiveAttentionGuidedGradcam"),
             "gridtd": ("ExplainImgCaptioningGridTDGradient", "ExplainImgCaptioningGridTDGradientTimesInput",
                        "ExplainImgCaptioningGridTDGuidedGradcam")}[kind]
    for cname, mode in zip(names, ("gradient", "input_x_gradient", "guided_backprop")):
        ex = getattr(EX, cname)(spec, None, None, max_caption_length=6)
        ex._forward_beam_search((None, X), cap)
        rel = ex._explain_sentence()
        assert len(rel) == len(cap) - 1 and rel[0].shape == (1, 4, 4, D)
        for i, d in enumerate(rel):
            dref = o.backward(i + 1)
            assert rel_l1(d, dref) < TOL
            img = ex._explain_CNN(X, d)
            ref = C.gradient_analyze(layers, X, dref, mode)
            if mode == "guided_backprop":
                ref = ref * grad_cam(feat, dref[0], L, D, upscale=4)[None, ..., None]
            assert img.shape == X.shape
            assert rel_l1(img, ref) < TOL, (cname, i, rel_l1(img, ref))
        one = ex._lstm_decoder_backward(2)
        assert rel_l1(one, o.backward(2)) < TOL
        np.testing.assert_allclose(ex.r_words, o.r_words, rtol=1e-3, atol=1e-7)
        with pytest.raises(NotImplementedError):
            ex._lstm_decoder_backward(len(cap) + 1)


def test_pyramid_expand_properties():
    """The scipy restatement of skimage's pyramid_expand: shape x16, preserves a constant image, smooth and bounded."""
    from lrp_imagecaptioning_amd.postprocess import pyramid_expand
    out = pyramid_expand(np.full((14, 14), 3.0), upscale=16, sigma=20)
    assert out.shape == (224, 224)
    np.testing.assert_allclose(out, 3.0, rtol=1e-12)
    rs = np.random.RandomState(0)
    a = rs.standard_normal((14, 14))
    b = pyramid_expand(a, upscale=16, sigma=20)
    assert a.min() - 1e-9 <= b.min() and b.max() <= a.max() + 1e-9
