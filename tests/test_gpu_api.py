"""-m gpu: the reference-facing Python surface (explainer classes, analyzer, harness,
LRP-inference driver) end to end against the CPU oracles."""
import numpy as np
import pytest

from conftest import rel_l1
from gpu_util import report
from lrp_imagecaptioning_amd.synthetic import adaptive_weights, vgg_weights
from oracle import cnn_lrp_ref as C
from oracle.decoder_ref import AdaptiveOracle

pytestmark = pytest.mark.gpu
TOL = 1e-4
CFG = [("c1", 3, 8, False), ("c2", 8, 8, True), ("c3", 8, 16, True), ("c4", 16, 16, False)]
HW, L, D, H, V = 16, 16, 16, 32, 40


def _weights(seed=0):
    rs = np.random.RandomState(seed)
    w = vgg_weights(rs, CFG, bias_std=0.3)
    w.update(adaptive_weights(rs, L, D, H, H, V))
    return w, rs


def _spec(w):
    from lrp_imagecaptioning_amd.explainers import CaptionModelSpec
    return CaptionModelSpec(w, img_encoder="vgg16", hidden_dim=H, embedding_dim=H, L=L, D=D, vocab_size=V,
                            cnn_cfg=CFG, img_hw=(HW, HW))


def _oracle(w, X, cap):
    layers = C.vgg_layers(w, CFG)
    feat = C.forward(layers, X).astype(np.float32)
    o = AdaptiveOracle(w, L, D, H, H)
    o.forward(feat, cap)
    return layers, o


def test_explainer_protocol_matches_oracle():
    from lrp_imagecaptioning_amd.explainers import ExplainImgCaptioningAdaptiveAttention
    w, rs = _weights()
    X = rs.uniform(-120, 130, size=(1, HW, HW, 3)).astype(np.float32)
    cap = [7, 12, 33, 5, 1]
    ex = ExplainImgCaptioningAdaptiveAttention(_spec(w), None, None, max_caption_length=8)
    ex._forward_beam_search((None, X), cap)
    layers, o = _oracle(w, X, cap)
    assert ex.ht.shape == (len(cap) + 1, H) and ex.ht.dtype == np.float32
    assert ex.context.dtype == np.float64 and ex.caption_preds.shape == (len(cap), V)
    assert rel_l1(ex.ht, o.ht) < 1e-5 and rel_l1(ex.caption_preds, o.caption_preds) < 1e-5
    rel, att = ex._explain_sentence()
    assert len(rel) == len(cap) - 1 and att.shape == (len(cap) - 1, L)
    worst = 0.0
    for i, R in enumerate(rel):
        assert R.shape == (1, 4, 4, D) and R.dtype == np.float32
        Rref, _ = o.explain(i + 1)
        img = ex._explain_CNN(X, R)
        ref = C.analyze(layers, X, Rref)
        assert img.shape == X.shape
        worst = max(worst, rel_l1(img, ref))
    report("api_protocol", max_rel_l1=worst)
    assert worst < TOL
    np.testing.assert_allclose(ex.r_words, o.r_words, rtol=1e-4, atol=1e-8)
    R1, a1 = ex._explain_lstm_single_word_sequence(2)
    assert rel_l1(R1, o.explain(2)[0]) < TOL
    np.testing.assert_allclose(ex.r_words, o.r_words, rtol=1e-4, atol=1e-8)
    R2, _ = ex._explain_lstm_single_word(2)
    assert rel_l1(R2, o.explain_single_step(2)[0]) < TOL
    with pytest.raises(NotImplementedError):
        ex._explain_lstm_single_word_sequence(len(cap) + 1)


def test_explain_batch_and_beam_search():
    from lrp_imagecaptioning_amd.explainers import ExplainImgCaptioningAdaptiveAttention
    w, rs = _weights(3)
    X = rs.uniform(-120, 130, size=(2, HW, HW, 3)).astype(np.float32)
    caps = [[9, 4, 1], [3, 17, 25, 30, 1]]
    ex = ExplainImgCaptioningAdaptiveAttention(_spec(w), None, None, max_caption_length=6, max_images=2)
    out, pairs, att, rw, _ = ex.explain_batch(X, caps)
    assert pairs == [(0, 1), (0, 2), (1, 1), (1, 2), (1, 3), (1, 4)]
    out = out.cpu().numpy()
    worst = 0.0
    for b in range(2):
        layers, o = _oracle(w, X[b:b + 1], caps[b])
        for j, (bb, t) in enumerate(pairs):
            if bb == b:
                worst = max(worst, rel_l1(out[j], C.analyze(layers, X[b:b + 1], o.explain(t)[0])[0]))
    assert worst < TOL
    beams = ex._beam_search((None, X[:1]), beam_size=3)
    assert len(beams) == 3 and all(b[-1] == 1 for b in beams)
    # greedy check of the best beam's first word against the oracle's step-0 logits
    _, o = _oracle(w, X[:1], [5, 1])
    first_words = {b[0] for b in beams}
    assert int(np.argmax(o.caption_preds[0])) + 1 in first_words


def test_analyzer_interface():
    from lrp_imagecaptioning_amd.analyzer import ImageModelSpec, LRPSequentialPresetA
    w, rs = _weights(5)
    with pytest.raises(ValueError):
        LRPSequentialPresetA(ImageModelSpec(w, CFG, (HW, HW)), epsilon=0.01, neuron_selection_mode="bogus")
    an = LRPSequentialPresetA(ImageModelSpec(w, CFG, (HW, HW)), epsilon=0.01, neuron_selection_mode="replace", max_batch=2)
    X = rs.uniform(-120, 130, size=(3, HW, HW, 3)).astype(np.float32)
    layers = C.vgg_layers(w, CFG)
    feat = C.forward(layers, X)
    R = (rs.standard_normal(feat.shape) * feat).astype(np.float32)
    out = an.analyze([X, R])                       # N=3 with max_batch=2: two chunks
    ref = C.analyze(layers, X, R)
    assert out.shape == X.shape and out.dtype == np.float32
    assert max(rel_l1(out[i], ref[i]) for i in range(3)) < TOL
    with pytest.raises(ValueError):
        an.analyze([X, R], neuron_selection=3)


def test_lrp_inference_layer_matches_reference_loop():
    from lrp_imagecaptioning_amd.explainers import (CaptionPreprocessorStub, DatasetProviderStub,
                                                    ExplainImgCaptioningAdaptiveAttention)
    from lrp_imagecaptioning_amd.lrp_inference import LRPInferenceLayerAdaptive
    from lrp_imagecaptioning_amd.postprocess import lrp_inference_score
    w, rs = _weights(8)
    B, T = 2, 5
    X = rs.uniform(-120, 130, size=(B, HW, HW, 3)).astype(np.float32)
    y = rs.standard_normal((B, T, V))
    y[0, 3, 0] = 50.0                              # image 0: EOS (id 1 = column 0) at position 3
    word_of = {i: "w%d" % i for i in range(1, V + 1)}
    stop = {"w%d" % (int(np.argmax(y[1, 1])) + 1)}  # make image 1 / position 1 a stop word
    prov = DatasetProviderStub(CaptionPreprocessorStub(2, 1, word_of))
    ex = ExplainImgCaptioningAdaptiveAttention(_spec(w), None, prov, max_caption_length=T, max_images=B)
    for mode in ("mean", "pos_mean", "quantile"):
        layer = LRPInferenceLayerAdaptive(ex, mode, stop_words=stop)
        got = layer.call([None, X, y])
        assert got.shape == y.shape
        want = np.zeros(y.shape)
        for b in range(B):                          # the reference loop, model.py:1657-1689, on the oracles
            cap = list(np.argmax(y[b], axis=-1) + 1)
            eos = cap.index(1) if 1 in cap else None
            full = [int(c) for c in (cap[:eos + 1] if eos is not None else cap + [1])]
            layers, o = _oracle(w, X[b:b + 1], full)
            for i in range(T):
                if word_of[int(cap[i])] in stop:
                    continue
                if cap[i] == 1:
                    break
                rel = C.analyze(layers, X[b:b + 1], o.explain(i + 1)[0])
                if cap[i] < V:
                    want[b, i, cap[i]] = lrp_inference_score(rel, mode)
        np.testing.assert_allclose(got, 1 + want, rtol=2e-3, atol=2e-5)
        assert (got[0, 3:] == 1).all()              # nothing after EOS


def test_harness_returns_arrays():
    from lrp_imagecaptioning_amd.explainers import ExplainImgCaptioningAdaptiveAttention
    from lrp_imagecaptioning_amd.harness import Explainer
    w, rs = _weights(2)
    spec = _spec(w)
    ex = ExplainImgCaptioningAdaptiveAttention(spec, None, None, max_caption_length=6)
    hz = Explainer(spec, None, ex, 6, beam_size=2)
    X = (2, rs.uniform(-120, 130, size=(1, HW, HW, 3)).astype(np.float32))
    caps = hz._predict_caption(X)
    res = hz._explain_captions(X, caps)
    n = len(caps) - 1
    assert res["relevance"].shape == (n, HW, HW, 3) and res["heatmaps"].shape == (n, HW, HW, 3)
    assert res["attention"].shape == (n, L)
    assert np.isfinite(res["relevance"]).all()
    from lrp_imagecaptioning_amd.postprocess import heatmap, postprocess
    for i in range(n):                                   # device-rendered heat-maps == the host rendering of the harness
        want = heatmap(postprocess(res["relevance"][i:i + 1], "BGRtoRGB", False))[0]
        assert (np.abs(res["heatmaps"][i] - want).max(axis=-1) > 0.02).mean() < 2e-3
    one = hz._explain_single_word(X, caps, None, 1)
    assert one["heatmap"].shape == (HW, HW, 3) and one["heatmap"].max() <= 255


def test_gridtd_explainer_class():
    from lrp_imagecaptioning_amd.explainers import CaptionModelSpec, ExplainImgCaptioningGridTDModel
    from lrp_imagecaptioning_amd.synthetic import gridtd_weights
    from oracle.decoder_ref import GridTDOracle
    rs = np.random.RandomState(4)
    w = vgg_weights(rs, CFG, bias_std=0.3)
    w.update(gridtd_weights(rs, L, D, H, H, V))
    spec = CaptionModelSpec(w, img_encoder="vgg16", hidden_dim=H, embedding_dim=H, L=L, D=D, vocab_size=V, cnn_cfg=CFG,
                            img_hw=(HW, HW))
    X = rs.uniform(-120, 130, size=(1, HW, HW, 3)).astype(np.float32)
    cap = [11, 4, 29, 1]
    ex = ExplainImgCaptioningGridTDModel(spec, None, None, max_caption_length=6)
    ex._forward_beam_search((None, X), cap)
    layers = C.vgg_layers(w, CFG)
    o = GridTDOracle(w, L, D, H, H)
    o.forward(C.forward(layers, X).astype(np.float32), cap)
    assert ex.h2t.shape == (len(cap) + 1, H) and ex.h2t.dtype == np.float64
    assert rel_l1(ex.h2t, o.h2t) < 1e-5 and rel_l1(ex.x2t, o.x2t) < 1e-5
    rel, att = ex._explain_sentence()
    worst = 0.0
    for i, R in enumerate(rel):
        img = ex._explain_CNN(X, R)
        worst = max(worst, rel_l1(img, C.analyze(layers, X, o.explain(i + 1)[0])))
    assert worst < TOL
    np.testing.assert_allclose(ex.r_words, o.r_words, rtol=1e-4, atol=1e-9)
    with pytest.raises(NotImplementedError):
        ex._explain_lstm_single_word(1)


def test_lrp_inference_layer_gridtd():
    """model.py:2013-2062 over the grid-TD engine; and the class refuses an adaptive explainer."""
    from lrp_imagecaptioning_amd.explainers import (CaptionModelSpec, ExplainImgCaptioningAdaptiveAttention,
                                                    ExplainImgCaptioningGridTDModel)
    from lrp_imagecaptioning_amd.lrp_inference import LRPInferenceLayergridTD
    from lrp_imagecaptioning_amd.postprocess import lrp_inference_score
    from lrp_imagecaptioning_amd.synthetic import gridtd_weights
    from oracle.decoder_ref import GridTDOracle
    rs = np.random.RandomState(12)
    w = vgg_weights(rs, CFG, bias_std=0.3)
    w.update(gridtd_weights(rs, L, D, H, H, V))
    spec = CaptionModelSpec(w, img_encoder="vgg16", hidden_dim=H, embedding_dim=H, L=L, D=D, vocab_size=V, cnn_cfg=CFG,
                            img_hw=(HW, HW))
    B, T = 2, 4
    X = rs.uniform(-120, 130, size=(B, HW, HW, 3)).astype(np.float32)
    y = rs.standard_normal((B, T, V))
    y[1, 2, 0] = 50.0
    ex = ExplainImgCaptioningGridTDModel(spec, None, None, max_caption_length=T, max_images=B)
    got = LRPInferenceLayergridTD(ex, "pos_mean").call([None, X, y])
    layers = C.vgg_layers(w, CFG)
    want = np.zeros(y.shape)
    for b in range(B):
        cap = list(np.argmax(y[b], axis=-1) + 1)
        full = [int(c) for c in (cap[:cap.index(1) + 1] if 1 in cap else cap + [1])]
        o = GridTDOracle(w, L, D, H, H)
        o.forward(C.forward(layers, X[b:b + 1]).astype(np.float32), full)
        for i in range(T):
            if cap[i] == 1:
                break
            if cap[i] < V:
                want[b, i, cap[i]] = lrp_inference_score(C.analyze(layers, X[b:b + 1], o.explain(i + 1)[0]), "pos_mean")
    np.testing.assert_allclose(got, 1 + want, rtol=2e-3, atol=2e-5)
    w2, _ = _weights(8)
    with pytest.raises(ValueError):
        LRPInferenceLayergridTD(ExplainImgCaptioningAdaptiveAttention(_spec(w2), None, None, max_caption_length=T), "mean")


@pytest.mark.parametrize("kind", ["adaptive", "gridtd"])
def test_incremental_beam_search_equals_replay(kind):
    """lrp_decoder_gen_begin / _gen_step (one decoder step per search step, state re-parented on the device) against
    the replay-based search (every step re-runs the partial captions, like the reference's predict_on_batch loop), and
    beam 1 against greedy decoding on the oracle's forward."""
    import lrp_imagecaptioning_amd.explainers as EX
    from lrp_imagecaptioning_amd.synthetic import gridtd_weights
    from oracle.decoder_ref import AdaptiveOracle, GridTDOracle
    rs = np.random.RandomState(21)
    w = vgg_weights(rs, CFG, bias_std=0.3)
    if kind == "adaptive":
        from lrp_imagecaptioning_amd.synthetic import adaptive_weights
        w.update(adaptive_weights(rs, L, D, H, H, V))
        cls, orc = EX.ExplainImgCaptioningAdaptiveAttention, AdaptiveOracle
    else:
        w.update(gridtd_weights(rs, L, D, H, H, V))
        cls, orc = EX.ExplainImgCaptioningGridTDModel, GridTDOracle
    spec = EX.CaptionModelSpec(w, img_encoder="vgg16", hidden_dim=H, embedding_dim=H, L=L, D=D, vocab_size=V, cnn_cfg=CFG,
                               img_hw=(HW, HW))
    X = rs.uniform(-120, 130, size=(2, HW, HW, 3)).astype(np.float32)
    ex = cls(spec, None, None, max_caption_length=7, max_images=4)
    for b in range(2):
        for beam in (1, 3, 4):
            got = ex._beam_search((None, X[b:b + 1]), beam_size=beam)
            want = ex._beam_search_replay((None, X[b:b + 1]), beam_size=beam)
            assert got == want, (kind, b, beam, got, want)
        # greedy: feed the arg-max word back, on the float64/float32 oracle
        layers = C.vgg_layers(w, CFG)
        feat = C.forward(layers, X[b:b + 1]).astype(np.float32)
        words = []
        for s in range(7):
            o = orc(w, L, D, H, H)
            o.forward(feat, words + [1])
            nxt = int(np.argmax(o.caption_preds[s])) + 1
            if nxt == 1:
                break
            words.append(nxt)
        g1 = ex._beam_search((None, X[b:b + 1]), beam_size=1)[0]
        assert g1[:len(words)] == words[:len(g1) - 1] or g1 == words + [1], (g1, words)
    # several images at once (inference.py:178-253): with max_images = 8 and beam 3 both images share every decoder step
    ex8 = cls(spec, None, None, max_caption_length=7, max_images=8)
    both = ex8._beam_search((None, X), beam_size=3)
    assert both == [ex._beam_search((None, X[b:b + 1]), beam_size=3) for b in range(2)]


def test_generation_api_errors():
    """State / argument errors of the incremental decoding entry points."""
    from lrp_imagecaptioning_amd.explainers import ExplainImgCaptioningAdaptiveAttention
    w, rs = _weights(5)
    ex = ExplainImgCaptioningAdaptiveAttention(_spec(w), None, None, max_caption_length=4, max_images=2)
    eng = ex._engine
    X = rs.uniform(-120, 130, size=(2, HW, HW, 3)).astype(np.float32)
    eng.encode_images(X)
    eng._gen_rows = 2
    with pytest.raises(RuntimeError):
        eng.gen_step(0)                                   # gen_begin has not run
    eng.gen_begin(2)
    lg = eng.gen_step(0)
    assert tuple(lg.shape) == (2, V) and np.isfinite(lg.cpu().numpy()).all()
    with pytest.raises(ValueError):
        eng.gen_step(1, [0, 5], [3, 3])                   # parent row out of range
    with pytest.raises(ValueError):
        eng.gen_step(1, [0, 1], [3, V + 1])               # word id out of range
    with pytest.raises(NotImplementedError):
        eng.gen_step(99, [0, 1], [3, 3])                  # step beyond max_caption_len (LRP_ERR_RANGE)
    with pytest.raises(RuntimeError):
        eng.decoder_explain([0], [1])                     # a search scratch is not a caption replay
    # re-parenting: both rows carry the SAME image (rows of one search share the features of their image); after
    # different first words their states differ, and swapping the parents swaps the logits of the next step
    feat = eng.get_features()[:1]
    eng.set_features(feat.expand(2, -1, -1).contiguous())

    def two_steps(parents):
        eng.gen_begin(2)
        eng.gen_step(0)
        eng.gen_step(1, [0, 1], [7, 9])
        return eng.gen_step(2, parents, [4, 4]).cpu().numpy()
    a, b = two_steps([0, 1]), two_steps([1, 0])
    assert np.abs(a[0] - a[1]).max() > 1e-6
    np.testing.assert_allclose(a[0], b[1], rtol=1e-12)
    np.testing.assert_allclose(a[1], b[0], rtol=1e-12)


def test_vgg19_topology():
    """'vgg19' (config.py:36-37, explain_image.py:17-20): the 16-conv topology up to block5_conv4, here at reduced
    width, through the explainer protocol against the oracle."""
    from lrp_imagecaptioning_amd.explainers import CaptionModelSpec, ExplainImgCaptioningAdaptiveAttention
    from lrp_imagecaptioning_amd.harness import Explainer
    from lrp_imagecaptioning_amd.synthetic import VGG19_CFG
    cfg = [(n, 3 if i == 0 else max(8, ci // 16), max(8, co // 16), p) for i, (n, ci, co, p) in enumerate(VGG19_CFG)]
    hw, Lv, Dv = 32, 4, cfg[-1][2]
    rs = np.random.RandomState(9)
    w = vgg_weights(rs, cfg, bias_std=0.2)
    w.update(adaptive_weights(rs, Lv, Dv, H, H, V))
    spec = CaptionModelSpec(w, img_encoder="vgg19", hidden_dim=H, embedding_dim=H, L=Lv, D=Dv, vocab_size=V, cnn_cfg=cfg,
                            img_hw=(hw, hw))
    assert len(CaptionModelSpec(w, img_encoder="vgg19", L=196, D=512).cnn_cfg) == 16
    ex = ExplainImgCaptioningAdaptiveAttention(spec, None, None, max_caption_length=6)
    assert Explainer(spec, None, ex, 6, beam_size=2)._color_conversion == "BGRtoRGB"
    X = rs.uniform(-120, 130, size=(1, hw, hw, 3)).astype(np.float32)
    cap = [7, 12, 33, 1]
    ex._forward_beam_search((None, X), cap)
    layers = C.vgg_layers(w, cfg)
    o = AdaptiveOracle(w, Lv, Dv, H, H)
    o.forward(C.forward(layers, X).astype(np.float32), cap)
    worst = 0.0
    for t in (1, 3):
        R, _ = ex._explain_lstm_single_word_sequence(t)
        Rref, _ = o.explain(t)
        worst = max(worst, rel_l1(ex._explain_CNN(X, R), C.analyze(layers, X, Rref)))
    report("api_vgg19", max_rel_l1=worst)
    assert worst < TOL, worst


def test_pipeline_two_handles_identical_to_one():
    """LRPPipeline: batches alternating over two handles / streams give exactly what one handle gives."""
    from lrp_imagecaptioning_amd.engine import LRPEngine
    from lrp_imagecaptioning_amd.pipeline import LRPPipeline
    import torch
    w, rs = _weights(3)
    kw = dict(decoder="adaptive", cnn_cfg=CFG, img_hw=(HW, HW), L=L, D=D, H=H, E=H, V=V, max_images=2, max_tokens=8, max_caption_len=5)
    pipe = LRPPipeline(2, **kw)
    pipe.set_weights(w)
    one = LRPEngine(**kw)
    one.set_weights(w)
    caps = [[5, 9, 17, 1], [8, 3, 1]]
    idx, tpos = [0, 0, 0, 1, 1], [1, 2, 3, 1, 2]
    batches = [torch.as_tensor(rs.uniform(-120, 130, size=(2, HW, HW, 3)).astype(np.float32)).cuda() for _ in range(5)]
    got = [pipe.explain_batch(X, caps, idx, tpos)[0] for X in batches]
    pipe.synchronize()
    for X, g in zip(batches, got):
        one.encode_images(X)
        one.decoder_forward(caps)
        want = one.explain_tokens(idx, tpos)[0]
        assert torch.equal(g, want)


def test_explicit_device_and_allocation_failure_are_statuses():
    """LRPEngine(device=k) is a public parameter: every ABI entry makes the handle's device current for the call and
    restores the caller's (csrc/engine.hip with_handle); a workspace that cannot be allocated comes back as
    LRP_ERR_NOMEM -> MemoryError, never as an abort, and leaves the process usable."""
    import torch
    from lrp_imagecaptioning_amd.engine import LRPEngine
    w, rs = _weights(5)
    kw = dict(decoder="adaptive", cnn_cfg=CFG, img_hw=(HW, HW), L=L, D=D, H=H, E=H, V=V, max_images=1, max_caption_len=5)
    with pytest.raises(MemoryError):
        LRPEngine(max_tokens=2 ** 31 - 1, **kw)               # 2^31 x (16 x 16 x 8) floats: hipMalloc must fail cleanly
    eng = LRPEngine(max_tokens=4, device=0, **kw)
    assert eng.device == torch.device("cuda", 0)
    eng.set_weights(w)
    X = rs.uniform(-120, 130, size=(1, HW, HW, 3)).astype(np.float32)
    before = torch.cuda.current_device()
    eng.encode_images(X)
    eng.decoder_forward([[5, 9, 1]])
    out = eng.explain_tokens([0, 0], [1, 2])[0]
    assert torch.cuda.current_device() == before and bool(torch.isfinite(out).all())
    if torch.cuda.device_count() > 1:                          # several GPUs in one process: the handle follows cfg.device
        e1 = LRPEngine(max_tokens=4, device=1, **kw)
        e1.set_weights(w)
        with torch.cuda.device(1):
            X1 = torch.as_tensor(X).cuda()
        e1.encode_images(X1)                                   # current device stays 0 in this thread
        e1.decoder_forward([[5, 9, 1]])
        o1 = e1.explain_tokens([0, 0], [1, 2])[0]
        assert o1.device.index == 1 and torch.equal(o1.cpu(), out.cpu())
        assert torch.cuda.current_device() == before


@pytest.mark.parametrize("kind", ["adaptive", "gridtd"])
def test_weights_set_from_device_match_host_set(kind):
    """lrp_set_weight_dev (the multi-GPU start-up path: the bundle arrives in HBM over RCCL and is packed by device
    kernels, no host round trip) must leave the handle in exactly the state lrp_set_weight does: heat-maps, decoder
    LRP, gradient baselines and the fine-tune step's first gradients bit for bit — also when some weights come from the
    host and some from the device."""
    import torch
    from lrp_imagecaptioning_amd.engine import LRPEngine
    from lrp_imagecaptioning_amd.synthetic import gridtd_weights
    rs = np.random.RandomState(12)
    w = vgg_weights(rs, CFG, bias_std=0.3)
    w.update((adaptive_weights if kind == "adaptive" else gridtd_weights)(rs, L, D, H, H, V))
    kw = dict(decoder=kind, cnn_cfg=CFG, img_hw=(HW, HW), L=L, D=D, H=H, E=H, V=V, max_images=2, max_tokens=6, max_caption_len=5)
    X = rs.uniform(-120, 130, size=(2, HW, HW, 3)).astype(np.float32)
    caps = [[5, 9, 17, 1], [8, 3, 1]]
    idx, tpos = [0, 0, 0, 1, 1], [1, 2, 3, 1, 2]

    def run(eng):
        eng.encode_images(X)
        eng.decoder_forward(caps)
        out, R, att, rw = eng.explain_tokens(idx, tpos, want_R_feat=True, want_attention=True, want_r_words=True)
        d, drw = eng.decoder_gradient(idx, tpos)
        g = eng.cnn_walk(idx, d, "gradient")
        return [t.clone() for t in (out, R, att, rw, d, drw, g)]

    host = LRPEngine(**kw)
    host.set_weights(w)
    want = run(host)
    wd = {k: torch.as_tensor(v).cuda() for k, v in w.items()}
    dev = LRPEngine(**kw)
    dev.set_weights_from_device(wd)
    got = run(dev)
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    # mixed: decoder from the host, encoder from the device, one weight overwritten by the other route afterwards
    mix = LRPEngine(**kw)
    mix.set_weights({k: v for k, v in w.items() if not k.startswith("c")})
    mix.set_weights_from_device({k: v for k, v in wd.items() if k.startswith("c")})
    mix.set_weights_from_device({"output_W": wd["output_W"]})
    mix.set_weights({"c2_W": w["c2_W"]})
    for a, b in zip(run(mix), want):
        assert torch.equal(a, b)
    # a second device set with other values really replaces the operands
    w2 = {k: (v * 1.25).astype(np.float32) for k, v in w.items()}
    dev.set_weights_from_device({k: torch.as_tensor(v).cuda() for k, v in w2.items()})
    host.set_weights(w2)
    for a, b in zip(run(dev), run(host)):
        assert torch.equal(a, b)
    # the fine-tune step starts from device-set weights too
    fresh_h, fresh_d = LRPEngine(**kw), LRPEngine(**kw)
    fresh_h.set_weights(w)
    fresh_d.set_weights_from_device(wd)
    cap_in = np.array([[1, 4, 8, 16], [1, 7, 2, 0]], dtype=np.int32)
    y = np.array([[4, 8, 16, 0], [7, 2, 0, -1]], dtype=np.int32)
    lw = (1 + rs.uniform(0, 1, size=(2, 4, V))).astype(np.float32)
    res = []
    for e in (fresh_h, fresh_d):
        e.train_begin(lr=1e-3)
        e.encode_images(X)
        g, l = e.train_step(cap_in, y, lw)
        e.train_apply(g)
        e.encode_images(X)
        g2, l2 = e.train_step(cap_in, y, lw)
        res.append((g.clone(), l.clone(), g2.clone(), l2.clone()))
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a, b)


def test_pipeline_result_consumed_on_the_callers_stream():
    """explain_batch(out=None) allocates the result under the slot's stream; a consumer on the caller's stream only needs
    `wait(slot)` — an event wait, not a device-wide synchronise — and gets complete heat-maps."""
    from lrp_imagecaptioning_amd.engine import LRPEngine
    from lrp_imagecaptioning_amd.pipeline import LRPPipeline
    import torch
    w, rs = _weights(4)
    kw = dict(decoder="adaptive", cnn_cfg=CFG, img_hw=(HW, HW), L=L, D=D, H=H, E=H, V=V, max_images=2, max_tokens=8, max_caption_len=5)
    pipe = LRPPipeline(2, **kw)
    pipe.set_weights(w)
    one = LRPEngine(**kw)
    one.set_weights(w)
    caps = [[5, 9, 17, 1], [8, 3, 1]]
    idx, tpos = [0, 0, 0, 1, 1], [1, 2, 3, 1, 2]
    sums, wants = [], []
    for i in range(6):
        X = torch.as_tensor(rs.uniform(-120, 130, size=(2, HW, HW, 3)).astype(np.float32)).cuda()
        out, slot = pipe.explain_batch(X, caps, idx, tpos)
        pipe.wait(slot)
        sums.append(out.double().abs().sum())               # consumer on the caller's (default) stream, no synchronize()
        del out
        one.encode_images(X)
        one.decoder_forward(caps)
        wants.append(one.explain_tokens(idx, tpos)[0].double().abs().sum())
    torch.cuda.synchronize()
    for a, b in zip(sums, wants):
        assert float(a) == float(b)


def test_log_softmax_topk_on_device_matches_numpy():
    """lrp_op_log_softmax_topk == `_log_softmax` + argpartition top-k (explainers.py:45-48, :76-78), at a vocabulary-sized row."""
    import torch
    from lrp_imagecaptioning_amd.beam import topk_log_softmax
    from lrp_imagecaptioning_amd.engine import log_softmax_topk
    rs = np.random.RandomState(3)
    for rows, Vn, k in [(6, 10000, 3), (1, 37, 5), (9, 257, 1), (2, 40, 32)]:
        x = rs.standard_normal((rows, Vn)) * 4
        x[0, 5] = x[0, 7] = x[0].max() + 1.0                # an exact tie for the top: the lower column comes first
        ids, lp = log_softmax_topk(torch.as_tensor(x).cuda(), k)
        rid, rlp = topk_log_softmax(x, k)
        assert np.array_equal(ids.cpu().numpy(), rid)
        np.testing.assert_allclose(lp.cpu().numpy(), rlp, rtol=0, atol=1e-12)
    with pytest.raises(ValueError):
        log_softmax_topk(torch.zeros((2, 10), dtype=torch.float64, device="cuda"), 33)
