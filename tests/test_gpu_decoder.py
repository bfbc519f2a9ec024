"""-m gpu: decoder half (lrp_decoder_forward / lrp_decoder_explain through the C ABI)
against the golden vectors produced by the reference's own numpy code
(tests/golden/*.npz) and against the CPU oracle on fresh seeds."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, rel_l1
from gpu_util import report
from lrp_imagecaptioning_amd.synthetic import decoder_case
from oracle.decoder_ref import AdaptiveOracle

pytestmark = pytest.mark.gpu
TOL = 1e-4
TINY_CFG = [("c1", 3, 8, False)]


def _engine(L, D, H, V, B, ntok, Tm):
    from lrp_imagecaptioning_amd.engine import LRPEngine
    g = int(round(np.sqrt(L)))
    # the CNN is not exercised here: a 1-conv stub encoder whose output matches (L, D)
    return LRPEngine(decoder="adaptive", cnn_cfg=[("c1", 3, D, False)], img_hw=(g, g), L=L, D=D, H=H, E=H, V=V,
                     max_images=B, max_tokens=ntok, max_caption_len=Tm)


def _load(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    L, D, H, E, V, T = [int(x) for x in g["dims"]]
    if "feat" in g.files:
        w = {k[2:]: g[k] for k in g.files if k.startswith("w_")}
        feat, cap = g["feat"], [int(c) for c in g["caption"]]
    else:
        w, feat, cap = decoder_case("adaptive", int(g["seed"]), L, D, H, V, T)
    return g, w, feat, cap, (L, D, H, V, T)


STATE = {"ht": "ht", "ct": "ct", "gt": "gt", "it_act": "it_act", "ft_act": "ft_act", "context": "context",
         "attention": "attention", "st": "st", "beta": "beta", "c_hat": "c_hat",
         "image_features_before_act": "image_features_before_act", "average_img_feature": "average_img_feature",
         "global_img_feature_before_act": "global_img_feature_before_act",
         "total_static_img_feature": "total_static_img_feature"}


@pytest.mark.parametrize("name", ["adaptive_small_s0", "adaptive_small_s1", "adaptive_small_s2"])
def test_adaptive_small_matches_reference(name):
    g, w, feat, cap, (L, D, H, V, T) = _load(name)
    eng = _engine(L, D, H, V, 2, 2 * len(cap), len(cap) + 2)
    eng.set_weights({k: v for k, v in w.items()})
    eng.set_features(np.concatenate([feat.reshape(1, L, D), feat.reshape(1, L, D)[:, ::-1]]))   # image 1 = decoy
    eng.decoder_forward([cap, cap[-3:] if len(cap) > 3 else cap])
    n = len(cap)
    for gk, sk in STATE.items():
        got = eng.read_state(sk)[0].cpu().numpy()
        ref = np.asarray(g["state_" + gk], dtype=np.float64)
        ref = ref.reshape(-1, ref.shape[-1]) if ref.ndim > 1 else ref.reshape(1, -1)
        got = got[:ref.shape[0]].astype(np.float64)
        assert rel_l1(got, ref) < 1e-5, (gk, rel_l1(got, ref))
    preds = eng.read_state("caption_preds")[0, :n].cpu().numpy()
    assert rel_l1(preds, g["state_caption_preds"]) < 1e-5
    xt = eng.read_state("xt")[0, :n].cpu().numpy()
    assert rel_l1(xt, g["state_xt"]) < 1e-6
    toks = [int(t) for t in g["tokens"]]
    R, att, rw = eng.decoder_explain([0] * len(toks), toks)
    R, att, rw = R.cpu().numpy(), att.cpu().numpy(), rw.cpu().numpy()
    errs, rw_errs = [], []
    for j, t in enumerate(toks):
        ref = g["R_feat"][j].reshape(L, D)
        errs.append(rel_l1(R[j], ref))
        np.testing.assert_allclose(att[j], g["attention_t"][j], rtol=1e-4, atol=1e-7)
        want = g["r_words_t%d" % t]
        # r_words is a by-product (word relevances printed by the harness); the float32 forward chain
        # sums in a different order than numpy's BLAS and ill-conditioned captions (tiny cell states)
        # amplify that ~1e-7 noise, so it gets a relative-L1 bound rather than an element-wise one
        if len(want):
            rw_errs.append(rel_l1(rw[j, :len(want)], want) if np.abs(want).sum() else 0.0)
        assert (rw[j, len(want):] == 0).all()
    report("dec_" + name, max_rel_l1=max(errs), r_words_rel_l1=max(rw_errs) if rw_errs else 0.0)
    assert max(errs) < TOL, errs
    assert not rw_errs or max(rw_errs) < 1e-3, rw_errs
    if "R_feat_single" in g.files:
        R1, _, _ = eng.decoder_explain([0] * len(toks), toks, variant="single_step")
        R1 = R1.cpu().numpy()
        e1 = max(rel_l1(R1[j], g["R_feat_single"][j].reshape(L, D)) for j in range(len(toks)))
        assert e1 < TOL, e1


def test_adaptive_full_size_matches_reference():
    """L=196, D=H=E=512, V=2000: outputs of the reference's own code, tokens 1 and 10."""
    g, w, feat, cap, (L, D, H, V, T) = _load("adaptive_full_s0")
    eng = _engine(L, D, H, V, 1, 4, len(cap))
    eng.set_weights(w)
    eng.set_features(feat.reshape(1, L, D))
    eng.decoder_forward([cap])
    toks = [int(t) for t in g["tokens"]]
    R, att, rw = eng.decoder_explain([0] * len(toks), toks)
    R = R.cpu().numpy()
    errs = [rel_l1(R[j], g["R_feat"][j].reshape(L, D)) for j in range(len(toks))]
    report("dec_adaptive_full", max_rel_l1=max(errs))
    assert max(errs) < TOL, errs
    np.testing.assert_allclose(att.cpu().numpy(), g["attention_t"], rtol=1e-4, atol=1e-8)


def test_adaptive_batch_vs_oracle_fresh_seed():
    """Several images with ragged caption lengths in one batch, against the CPU oracle."""
    L, D, H, V = 16, 32, 32, 60
    eng = _engine(L, D, H, V, 3, 32, 8)
    w, _, _ = decoder_case("adaptive", 11, L, D, H, V, 3)
    eng.set_weights(w)
    rs = np.random.RandomState(5)
    feats, caps = [], []
    for T in (2, 7, 4):
        feats.append(np.maximum(rs.standard_normal((1, 4, 4, D)), 0).astype(np.float32))
        caps.append([int(c) for c in rs.randint(3, V + 1, size=T)] + [1])
    eng.set_features(np.concatenate(feats).reshape(3, L, D))
    eng.decoder_forward(caps)
    pairs = [(b, t) for b in range(3) for t in range(1, len(caps[b]))]
    R, _, _ = eng.decoder_explain([p[0] for p in pairs], [p[1] for p in pairs])
    R = R.cpu().numpy()
    errs = []
    for b in range(3):
        o = AdaptiveOracle(w, L, D, H, H)
        o.forward(feats[b], caps[b])
        for j, (bb, t) in enumerate(pairs):
            if bb == b:
                errs.append(rel_l1(R[j], o.explain(t)[0].reshape(L, D)))
    report("dec_batch_oracle", max_rel_l1=max(errs))
    assert max(errs) < TOL, errs


def test_out_of_range_token_raises_like_reference():
    L, D, H, V = 16, 32, 32, 60
    eng = _engine(L, D, H, V, 1, 4, 6)
    w, feat, cap = decoder_case("adaptive", 3, L, D, H, V, 3)
    eng.set_weights(w)
    eng.set_features(feat.reshape(1, L, D))
    with pytest.raises(RuntimeError):
        eng.decoder_explain([0], [1])                   # forward not run yet
    eng.decoder_forward([cap])
    with pytest.raises(NotImplementedError):            # E:538-539
        eng.decoder_explain([0], [len(cap) + 1])
    with pytest.raises(NotImplementedError):
        eng.decoder_explain([0], [0])


# ----------------------------------------------------------------------------------- grid-TD
def _engine_gtd(L, D, H, V, B, ntok, Tm):
    from lrp_imagecaptioning_amd.engine import LRPEngine
    g = int(round(np.sqrt(L)))
    return LRPEngine(decoder="gridtd", cnn_cfg=[("c1", 3, D, False)], img_hw=(g, g), L=L, D=D, H=H, E=H, V=V,
                     max_images=B, max_tokens=ntok, max_caption_len=Tm)


def _load_gtd(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    L, D, H, E, V, T = [int(x) for x in g["dims"]]
    if "feat" in g.files:
        w = {k[2:]: g[k] for k in g.files if k.startswith("w_")}
        feat, cap = g["feat"], [int(c) for c in g["caption"]]
    else:
        w, feat, cap = decoder_case("gridtd", int(g["seed"]), L, D, H, V, T)
    return g, w, feat, cap, (L, D, H, V, T)


GTD_STATE = ["h1t", "c1t", "g1t", "i1t_act", "f1t_act", "h2t", "c2t", "g2t", "i2t_act", "f2t_act", "context", "st",
             "beta", "context_hat", "attention"]


@pytest.mark.parametrize("name", ["gridtd_small_s0", "gridtd_small_s1", "gridtd_small_s2"])
def test_gridtd_small_matches_reference(name):
    g, w, feat, cap, (L, D, H, V, T) = _load_gtd(name)
    eng = _engine_gtd(L, D, H, V, 2, 2 * len(cap), len(cap) + 2)
    eng.set_weights(w)
    eng.set_features(np.concatenate([feat.reshape(1, L, D)[:, ::-1], feat.reshape(1, L, D)]))   # image 0 = decoy
    eng.decoder_forward([cap[-2:] if len(cap) > 2 else cap, cap])
    n = len(cap)
    for k in GTD_STATE:
        got = eng.read_state(k)[1].cpu().numpy()
        ref = np.asarray(g["state_" + k], dtype=np.float64)
        assert rel_l1(got[:ref.shape[0]], ref) < 1e-5, (k, rel_l1(got[:ref.shape[0]], ref))
    for k in ("x1t", "x2t", "caption_preds"):
        assert rel_l1(eng.read_state(k)[1, :n].cpu().numpy(), g["state_" + k]) < 1e-5, k
    assert rel_l1(eng.read_state("image_features_proj")[1].cpu().numpy(), g["state_image_features_proj_bm"]) < 1e-5
    toks = [int(t) for t in g["tokens"]]
    R, att, rw = eng.decoder_explain([1] * len(toks), toks)
    R, att, rw = R.cpu().numpy(), att.cpu().numpy(), rw.cpu().numpy()
    errs = []
    for j, t in enumerate(toks):
        errs.append(rel_l1(R[j], g["R_feat"][j].reshape(L, D)))
        np.testing.assert_allclose(att[j], g["attention_t"][j], rtol=1e-4, atol=1e-7)
        want = g["r_words_t%d" % t]
        np.testing.assert_allclose(rw[j, :len(want)], want, rtol=1e-4, atol=1e-8)
    report("dec_" + name, max_rel_l1=max(errs))
    assert max(errs) < TOL, errs
    with pytest.raises(NotImplementedError):
        eng.decoder_explain([1], [1], variant="single_step")


def test_gridtd_full_size_matches_reference():
    g, w, feat, cap, (L, D, H, V, T) = _load_gtd("gridtd_full_s0")
    eng = _engine_gtd(L, D, H, V, 1, 4, len(cap))
    eng.set_weights(w)
    eng.set_features(feat.reshape(1, L, D))
    eng.decoder_forward([cap])
    toks = [int(t) for t in g["tokens"]]
    R, att, _ = eng.decoder_explain([0] * len(toks), toks)
    R = R.cpu().numpy()
    errs = [rel_l1(R[j], g["R_feat"][j].reshape(L, D)) for j in range(len(toks))]
    report("dec_gridtd_full", max_rel_l1=max(errs))
    assert max(errs) < TOL, errs
