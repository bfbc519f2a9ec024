"""oracle/train_ref.py: pinned against the decoder oracle (itself pinned by reference outputs) and by finite differences."""
import numpy as np
import torch

from lrp_imagecaptioning_amd.synthetic import adaptive_weights, vgg_weights
from oracle import cnn_lrp_ref as C
from oracle import train_ref as T
from oracle.decoder_ref import AdaptiveOracle

CFG = [("c1", 3, 8, True), ("c2", 8, 16, True), ("c3", 16, 16, False)]
HW, L, D, H, V = 16, 16, 16, 16, 24


def _case(seed=0, B=2, Tn=4):
    rs = np.random.RandomState(seed)
    w = vgg_weights(rs, CFG, bias_std=0.3)
    w.update(adaptive_weights(rs, L, D, H, H, V))
    X = rs.uniform(-120, 130, size=(B, HW, HW, 3)).astype(np.float32) / 64
    caps = [[int(c) for c in rs.randint(3, V + 1, size=Tn - 1)] + [1] for _ in range(B)]
    cap_in = np.array([[2 - 1] + [c - 1 for c in cap[:-1]] for cap in caps])        # SOS then the shifted caption
    y = np.array([[c - 1 for c in cap] for cap in caps])
    y[1, -2:] = -1                                                                  # a padded tail
    lw = 1 + rs.uniform(0, 1, size=(B, Tn, V)) * (rs.uniform(size=(B, Tn, V)) < 0.2)
    return rs, w, X, caps, cap_in, y, lw


def test_logits_match_decoder_oracle():
    rs, w, X, caps, cap_in, y, lw = _case()
    _, _, _, _, logits = T.loss_and_grads(w, CFG, X, cap_in, y, lw)
    layers = C.vgg_layers(w, CFG)
    for b in range(len(caps)):
        o = AdaptiveOracle(w, L, D, H, H)
        o.forward(C.forward(layers, X[b:b + 1]).astype(np.float32), caps[b])
        np.testing.assert_allclose(logits[b], o.caption_preds, rtol=2e-4, atol=2e-5)   # (the oracle's chain is float32)


def test_gradients_by_central_differences():
    rs, w, X, caps, cap_in, y, lw = _case(1)
    B, Tn = cap_in.shape
    p = 0.5
    mk = lambda *s: (rs.uniform(size=s) >= p) / (1 - p)
    masks = {"image_features": mk(B, L, H), "global": mk(B, H), "output": mk(B, Tn, H),
             "lstm_in": mk(Tn, 4, B, 2 * H), "lstm_rec": mk(Tn, 4, B, H)}
    total, l1, l2, g, _ = T.loss_and_grads(w, CFG, X, cap_in, y, lw, masks)
    assert np.isclose(total, 0.5 * l1 + 0.5 * l2)
    w64 = {k: np.asarray(v, np.float64) for k, v in w.items()}
    for name in ("c1_W", "c2_b", "c3_W", "image_features_W", "global_b", "embedding", "lstm_Wi", "lstm_Wh", "lstm_b", "Wv",
                 "Wg", "V", "Wx", "Wh", "Ws", "output_W", "output_b"):
        flat = np.abs(g[name]).ravel()
        idx = np.unravel_index(int(np.argmax(flat)), g[name].shape)
        eps = 1e-5
        vals = []
        for sgn in (+1, -1):
            wp = dict(w64)
            wp[name] = w64[name].copy()
            wp[name][idx] += sgn * eps
            vals.append(T.loss_and_grads(wp, CFG, X, cap_in, y, lw, masks)[0])
        fd = (vals[0] - vals[1]) / (2 * eps)
        assert abs(fd - g[name][idx]) <= 1e-5 * max(1.0, abs(fd)) + 1e-8, (name, fd, g[name][idx])


def test_adam_clipvalue_step():
    rs = np.random.RandomState(0)
    p, g = rs.standard_normal(6), np.array([0.5, -0.5, 0.001, -0.002, 0.0, 0.02])
    m = v = np.zeros(6)
    p1, m1, v1 = T.adam_clipvalue_step(p, g, m, v, 1, lr=1e-3, clipvalue=0.01)
    gc = np.clip(g, -0.01, 0.01)
    np.testing.assert_allclose(m1, 0.1 * gc)
    np.testing.assert_allclose(v1, 0.001 * gc * gc)
    # first step of Adam moves every touched parameter by ~lr against the sign of its gradient
    lr_t = 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9)
    np.testing.assert_allclose(p1, p - lr_t * m1 / (np.sqrt(v1) + 1e-7))
    assert np.all(np.sign(p - p1)[g != 0] == np.sign(g)[g != 0]) and p1[4] == p[4]


def test_gridtd_hidden_states_match_decoder_oracle():
    """The grid-TD training graph against oracle/decoder_ref.GridTDOracle (pinned by reference outputs).  The explainer's
    replay takes its logits from h2 alone (E:1154) while the Keras model uses h2 + c_hat (M:816): compare h2 + c_hat."""
    from lrp_imagecaptioning_amd.synthetic import gridtd_weights
    from oracle.decoder_ref import GridTDOracle
    rs = np.random.RandomState(2)
    w = vgg_weights(rs, CFG, bias_std=0.3)
    w.update(gridtd_weights(rs, L, D, H, H, V))
    X = rs.uniform(-2, 2, size=(1, HW, HW, 3)).astype(np.float32)
    cap = [5, 9, 14, 1]
    cap_in = np.array([[1] + [c - 1 for c in cap[:-1]]])
    y = np.array([[c - 1 for c in cap]])
    _, _, _, g, logits = T.loss_and_grads(w, CFG, X, cap_in, y, np.ones((1, 4, V)), kind="gridtd")
    o = GridTDOracle(w, L, D, H, H)
    o.forward(C.forward(C.vgg_layers(w, CFG), X).astype(np.float32), cap)
    want = (o.h2t[1:] + o.context_hat[1:]) @ w["output_W"] + w["output_b"]
    np.testing.assert_allclose(logits[0], want, rtol=2e-4, atol=2e-5)
    assert set(g) == set(T.param_names(CFG, "gridtd")) and all(np.isfinite(v).all() for v in g.values())
