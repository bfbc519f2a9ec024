"""ORACLE — test infrastructure only (never imported by the product path).

CPU restatement (numpy) of the hand-written BPTT of the reference's gradient baselines
(models/explainers.py, cited as E:<line>):
    adaptive  _lstm_decoder_backward     E:780-832   (class E:667, forward E:690-778)
    grid-TD   _lstm_decoder_backward     E:1452-1532 (class E:1322, forward E:1344-1450)
The reference's backward is NOT the true gradient: attention weights, beta and the sentinel are treated as
constants, the visual-sentinel branch of c_hat gets no gradient, and d_context is taken un-scaled by (1 - beta)
in the adaptive model, whose relu mask of the global feature is also a scalar quirk (see E:827 below).  It is
restated with those simplifications; the forward is the LRP oracle's forward
(the same recurrences) plus the output-gate and tanh(g) activations the backward needs.

Pinned by tests/test_oracle_decoder.py against tests/golden/*_grad_*.npz, produced by running the reference's
own code (tests/golden/make_golden.py --only grad).
"""
import numpy as np
from scipy.special import expit

from .decoder_ref import AdaptiveOracle, GridTDOracle


def _o_gates(xs, hs, Wi, Wh, b, H):
    """o_act rows 1..T (row 0 = zeros like the reference's state arrays): sigmoid of the 4th gate block (E:673-688)."""
    rows = [np.zeros((1, H), dtype="float32")]
    for i in range(len(xs)):
        z = np.dot(xs[i:i + 1], Wi) + np.dot(hs[i:i + 1], Wh) + b
        rows.append(expit(z[:, 3 * H:]).astype("float32"))
    return np.vstack(rows)


def _lstm_cell_backward(dh, dc_next, c, c_prev, i, f, g, o):
    """One step of E:808-818 / E:1490-1500.  Returns (dc_prev, d_gates (4H))."""
    tc = np.tanh(c)
    d_o_act = dh * tc
    dc = dc_next + dh * o * (1.0 - tc ** 2)
    d_f_act = dc * c_prev
    dc_prev = dc * f
    d_i_act = dc * g
    d_g_act = dc * i
    d_i = d_i_act * i * (1 - i)
    d_f = d_f_act * f * (1 - f)
    d_o = d_o_act * o * (1 - o)
    d_g = d_g_act * (1 - g ** 2)
    return dc_prev, np.hstack((d_i, d_f, d_g, d_o))


class AdaptiveGradOracle(AdaptiveOracle):
    def forward(self, feat, caption, sos=2):
        super(AdaptiveGradOracle, self).forward(feat, caption, sos)
        w = self.w
        self.ot_act = _o_gates(self.xt, self.ht[:-1], w["lstm_Wi"], w["lstm_Wh"], w["lstm_b"], self.H)
        self.gt_act = np.tanh(self.gt)

    def backward(self, t):
        """E:780-832 -> (d_img_feature (1, sqrtL, sqrtL, D) float32; self.r_words (t,))."""
        w, H, E, L, D = self.w, self.H, self.E, self.L, self.D
        k = self.caption[t - 1] - 1
        f32 = lambda *s: np.zeros(s, dtype="float32")
        d_ht, d_ct = f32(t + 1, H), f32(t + 1, H)
        d_words = np.zeros((t, E))
        d_seed = w["output_W"][:, k].astype(np.float64)[None]            # E:801: d_caption_preds . W_out^T
        d_context = d_seed                                                # E:803-804 (c_hat -> context, unscaled)
        d_ht[t] = d_seed
        d_glob = np.zeros(E)
        d_V = f32(L, H)
        for l in range(L):                                                # E:807-808
            d_V[l] = d_context * self.attention[t, l]
        d_V[self.Vfeat <= 0] = 0                                          # E:809
        for i in range(t)[::-1]:                                          # E:810-826
            dc_prev, dg = _lstm_cell_backward(d_ht[i + 1], d_ct[i + 1], self.ct[i + 1], self.ct[i], self.it_act[i + 1],
                                              self.ft_act[i + 1], self.gt_act[i + 1], self.ot_act[i + 1])
            d_ct[i] = dc_prev
            dg = dg.astype("float32")[None]
            d_ht[i] = np.dot(dg, w["lstm_Wh"].T)
            d_xt = np.dot(dg, w["lstm_Wi"].T)[0]
            d_glob += d_xt[E:]
            d_words[i] = d_xt[:E]
        # E:827 `d_global_img_feature[self._global_img_feature[0]<=0] = 0`: the global feature is a 1-D vector here,
        # so `[0]` is its first ELEMENT and the mask is a scalar boolean — all of d_glob is zeroed when that one
        # element is not positive, nothing otherwise (the grid-TD class masks element-wise, E:1523)
        if self.glob.reshape(-1)[0] <= 0:
            d_glob[:] = 0
        d_avg = np.dot(d_glob, w["global_W"].T)                           # E:828
        out = f32(L, D)
        for l in range(L):                                                # E:829-831
            out[l] = 1.0 * d_avg / L
            out[l] += np.dot(d_V[l], w["image_features_W"].T)
        self.r_words = np.sum(d_words, axis=-1)
        s = int(np.sqrt(L))
        return out.reshape(1, s, s, D)


class GridTDGradOracle(GridTDOracle):
    def forward(self, feat, caption, sos=2):
        super(GridTDGradOracle, self).forward(feat, caption, sos)
        w, H = self.w, self.H
        self.o1t_act = _o_gates(self.x1t, self.h1t[:-1], w["td_Wi"], w["td_Wh"], w["td_b"], H)
        self.o2t_act = _o_gates(self.x2t, self.h2t[:-1], w["lang_Wi"], w["lang_Wh"], w["lang_b"], H)
        self.g1t_act, self.g2t_act = np.tanh(self.g1t), np.tanh(self.g2t)

    def backward(self, t):
        """E:1452-1532."""
        w, H, E, L, D = self.w, self.H, self.E, self.L, self.D
        k = self.caption[t - 1] - 1
        f32 = lambda *s: np.zeros(s, dtype="float32")
        d_h1t, d_c1t, d_h2t, d_c2t = f32(t + 1, H), f32(t + 1, H), f32(t + 1, H), f32(t + 1, H)
        d_V = np.zeros((L, H))
        d_words = np.zeros((t, E))
        d_glob = np.zeros((1, E))
        d_context_hat = np.zeros((t, H))
        d_seed = w["output_W"][:, k].astype(np.float64)[None]            # E:1484
        d_context_hat[t - 1] = d_seed
        d_h2t[t] = d_seed
        for i in range(t)[::-1]:
            dc_prev, dg2 = _lstm_cell_backward(d_h2t[i + 1], d_c2t[i + 1], self.c2t[i + 1], self.c2t[i], self.i2t_act[i + 1],
                                               self.f2t_act[i + 1], self.g2t_act[i + 1], self.o2t_act[i + 1])
            d_c2t[i] = dc_prev
            dg2 = dg2.astype("float32")[None]
            d_h2t[i] = np.dot(dg2, w["lang_Wh"].T)
            d_xt2 = np.dot(dg2, w["lang_Wi"].T)[0]
            d_context_hat[i] += d_xt2[:H]                                  # E:1502
            d_context = d_context_hat[i] * (1 - self.beta[i + 1][0])       # E:1503
            d_h1t[i + 1] += d_xt2[H:]                                      # E:1504
            dc_prev, dg1 = _lstm_cell_backward(d_h1t[i + 1], d_c1t[i + 1], self.c1t[i + 1], self.c1t[i], self.i1t_act[i + 1],
                                               self.f1t_act[i + 1], self.g1t_act[i + 1], self.o1t_act[i + 1])
            d_c1t[i] = dc_prev
            dg1 = dg1.astype("float32")[None]
            d_h1t[i] = np.dot(dg1, w["td_Wh"].T)
            d_xt1 = np.dot(dg1, w["td_Wi"].T)[0]
            d_glob += d_xt1[H:H + E]                                       # E:1518
            d_words[i] = d_xt1[H + E:]                                     # E:1519
            for l in range(L):                                             # E:1520-1521
                d_V[l] += d_context * self.attention[i + 1][l]
            d_h2t[i] += d_xt1[:H]                                          # E:1522
        d_glob[0][self.glob <= 0] = 0                                      # E:1523
        d_avg = np.dot(d_glob, w["global_W"].T)                            # E:1524
        d_V[self.Vfeat <= 0] = 0                                           # E:1525
        self.r_words = np.sum(d_words, axis=-1)
        out = f32(L, D)
        for l in range(L):                                                 # E:1527-1529
            out[l] = np.dot(d_V[l], w["image_features_W"].T)
            out[l] += d_avg[0] / L
        s = int(np.sqrt(L))
        return out.reshape(1, s, s, D)
