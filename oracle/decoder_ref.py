"""ORACLE — test infrastructure only (never imported by the product path).

CPU restatement (numpy) of the reference's decoder-half LRP:
    models/explainers.py  (cited below as E:<line>)
      _lstm_forward                        E:125-139
      _get_sign_stabilizer                 E:141-144
      _propagate_relevance_linear_lrp      E:156-165
      adaptive  _forward_beam_search       E:370-436
      adaptive  _explain_lstm_single_word  E:438-535
      adaptive  _explain_lstm_single_word_sequence   E:537-666
      grid-TD   _forward_beam_search       E:1092-1178
      grid-TD   _explain_lstm_single_word_sequence   E:1180-1321

It keeps the reference's *cost structure* on purpose (one dense (Din,Dout)
attribution matrix per rule call, identity weights for the element-wise
shares, a Python loop over the L image locations) because bench.py times it
as the `cpu_baseline` ("port").  It also keeps the reference's dtype mix:
float32 LSTM/attention chain, float64 from `context` on, float32 stores into
r_V / r_img_feature_input.

Pinned by tests/test_oracle_decoder.py against tests/golden/*.npz, which were
produced by running the reference's own code (tests/golden/make_golden.py).
"""
import numpy as np
from scipy.special import expit, softmax

LRP_EPS = 1e-7          # K.epsilon() bound as default arg at E:157


def stabilize(z, eps=LRP_EPS):
    """z + sign(z)*eps with sign(0)=+1  (E:141-144)."""
    s = np.ones(z.shape)
    s[z < 0] = -1
    return z + s * eps


def linear_lrp(r_out, x, z, weight, bias=None, bias_nb_units=1, bias_factor=0, eps=LRP_EPS):
    """The decoder rule (E:156-165).  weight (Din,Dout), x (Din,), z (Dout,),
    r_out (Dout,) or (1,Dout) -> (Din,).  bias_factor is 0 at every call site."""
    attribution = weight * x[:, None]
    if bias_factor:
        attribution = attribution + (bias_factor * 1.0 * bias[None, :]) / bias_nb_units
    return np.sum(attribution / stabilize(z, eps) * r_out, axis=1)


def lstm_step(x, h_prev, c_prev, Wi, Wh, b, H):
    """E:125-139; returns (h, c, g_preact, i_act, f_act)."""
    z = np.dot(x, Wi)
    z += np.dot(h_prev, Wh)
    z = z + b
    i = expit(z[:, :H])
    f = expit(z[:, H:2 * H])
    g = z[:, 2 * H:3 * H]
    o = expit(z[:, 3 * H:])
    c = f * c_prev + i * np.tanh(g)
    return o * np.tanh(c), c, g, i, f


def _gate_g_weight(Wi, Wh):
    """[Wi;Wh][:, 2H:3H]  (E:556-558)."""
    return np.vstack((np.split(Wi, 4, 1)[2], np.split(Wh, 4, 1)[2]))


class _Base(object):
    def __init__(self, weights, L, D, H, E):
        self.w = weights
        self.L, self.D, self.H, self.E = L, D, H, E

    def _static_image_part(self, feat):
        """E:375-388 / E:1097-1107 (shared prologue)."""
        w = self.w
        self.F = feat.reshape(self.L, self.D)
        self.if_pre = np.zeros((self.L, self.H))
        for i in range(self.L):
            self.if_pre[i] = np.dot(self.F[i], w["image_features_W"]) + w["image_features_b"]
        self.Vfeat = np.maximum(self.if_pre, 0)
        self.avg = np.mean(self.F, axis=0)
        self.glob_pre = np.dot(self.avg, w["global_W"]) + w["global_b"]
        self.glob = np.maximum(self.glob_pre, 0)

    def _image_tail(self, r_glob, r_V):
        """global-feature rule + loop over L (mean-pool share, image_features
        dense rule): E:634-659 / E:1301-1319.  r_V is the float32 (L,H) array."""
        w = self.w
        I_D = None
        r_avg = linear_lrp(r_glob, self.avg, self.glob_pre, w["global_W"])
        R = np.zeros((self.L, self.D), dtype="float32")
        for i in range(self.L):
            I_D = np.identity(self.D)
            R[i] = linear_lrp(r_avg, self.F[i] / self.L, self.avg, I_D)
            R[i] += linear_lrp(r_V[i], self.F[i], self.if_pre[i], w["image_features_W"])
        g = int(np.sqrt(self.L))
        return R.reshape(1, g, g, self.D)


class AdaptiveOracle(_Base):
    """ExplainImgCaptioningAdaptiveAttention (E:260-666) restated."""

    def forward(self, feat, caption, sos=2):
        """E:370-436.  caption = tokenizer ids (1-based), last one EOS."""
        w, H, E, L = self.w, self.H, self.E, self.L
        self.caption = list(caption)
        self._static_image_part(feat)
        self.static = np.dot(self.Vfeat, w["Wv"])
        z32 = lambda *s: np.zeros(s, dtype="float32")
        self.ht, self.ct, self.gt, self.it_act, self.ft_act = z32(1, H), z32(1, H), z32(1, H), z32(1, H), z32(1, H)
        self.context, self.st, self.c_hat = z32(1, H), z32(1, H), z32(1, H)
        self.attention, self.beta = z32(1, L), z32(1, 1)
        xs, preds = [], []
        for i in range(len(caption)):
            h_prev, c_prev = self.ht[-1], self.ct[-1]
            tok = (sos if i == 0 else caption[i - 1]) - 1
            x = np.hstack((w["embedding"][tok][None], self.glob.reshape(1, E)))
            h, c, g, ia, fa = lstm_step(x, h_prev, c_prev, w["lstm_Wi"], w["lstm_Wh"], w["lstm_b"], H)
            h_proj = np.dot(h, w["Wg"])
            att_pre = np.dot(np.tanh(h_proj + self.static, dtype="float32"), w["V"])
            att = softmax(att_pre, axis=0)
            s = np.tanh(c) * expit(np.dot(x, w["Wx"]) + np.dot(h_prev, w["Wh"]))
            z_s = np.dot(np.tanh(np.dot(s, w["Ws"]) + h_proj), w["V"])
            beta = softmax(np.concatenate((att_pre, z_s), axis=0), axis=0)[-1][0]
            ctx = np.sum(att * self.Vfeat, axis=0).reshape(1, H)
            c_hat = beta * s + (1 - beta) * ctx
            preds.append(np.dot(h + c_hat, w["output_W"]) + w["output_b"])
            xs.append(x)
            self.ht = np.vstack((self.ht, h))
            self.ct = np.vstack((self.ct, c))
            self.gt = np.vstack((self.gt, g))
            self.it_act = np.vstack((self.it_act, ia))
            self.ft_act = np.vstack((self.ft_act, fa))
            self.context = np.vstack((self.context, ctx))
            self.attention = np.vstack((self.attention, att.reshape(1, L)))
            self.st = np.vstack((self.st, s))
            self.beta = np.vstack((self.beta, beta.reshape(1, 1)))
            self.c_hat = np.vstack((self.c_hat, c_hat))
        self.xt = np.vstack(xs)
        self.caption_preds = np.vstack(preds)

    def _head(self, t):
        """Output layer + h/c_hat split + context/sentinel split (E:552-601)."""
        w, H = self.w, self.H
        k = self.caption[t - 1] - 1
        seed = np.zeros((1, self.caption_preds.shape[1]))
        seed[0, k] = self.caption_preds[t - 1, k]
        u = self.ht[t] + self.c_hat[t]
        I = np.identity
        r_u = linear_lrp(seed, u, self.caption_preds[t - 1], w["output_W"])
        r_h = linear_lrp(r_u, self.ht[t], u, I(H))
        r_chat = linear_lrp(r_u, self.c_hat[t], u, I(H))
        b = self.beta[t][0]
        r_ctx = linear_lrp(r_chat, (1 - b) * self.context[t], self.c_hat[t], I(H))
        r_s = linear_lrp(r_chat, b * self.st[t], self.c_hat[t], I(H))
        return r_h, r_ctx, r_s

    def _attention_sum(self, r_ctx, t):
        """r_V[i] = share(r_ctx, V_i * alpha_i, ctx)  (E:648-653)."""
        r_V = np.zeros((self.L, self.H), dtype="float32")
        for i in range(self.L):
            r_V[i] = linear_lrp(r_ctx, self.Vfeat[i] * self.attention[t][i], self.context[t], np.identity(self.H))
        return r_V

    def explain(self, t):
        """E:537-666.  Returns (R_feat (1,g,g,D) float32, attention_t (L,));
        sets self.r_words."""
        if t > len(self.xt):
            raise NotImplementedError("index out of range of captions")
        w, H, E = self.w, self.H, self.E
        W_g = _gate_g_weight(w["lstm_Wi"], w["lstm_Wh"])
        xh = np.hstack((self.xt[0:t], self.ht[0:t]))
        r_c = np.zeros((t + 1, H))
        r_h = np.zeros((t + 1, H))
        r_words = np.zeros((t, E))
        r_glob = np.zeros(E)
        r_h[t], r_ctx, r_s = self._head(t)
        r_c[t] = r_s
        I = np.identity
        for i in range(t)[::-1]:
            r_c[i + 1] += r_h[i + 1]
            r_g = linear_lrp(r_c[i + 1], self.it_act[i + 1] * np.tanh(self.gt[i + 1]), self.ct[i + 1], I(H))
            r_c[i] = linear_lrp(r_c[i + 1], self.ft_act[i + 1] * self.ct[i], self.ct[i + 1], I(H))
            r_xh = linear_lrp(r_g, xh[i], self.gt[i + 1], W_g)
            r_h[i] = r_xh[2 * E:]                 # '=' not '+='  (E:627)
            r_glob += r_xh[E:2 * E]
            r_words[i] = r_xh[:E]
        R = self._image_tail(r_glob, self._attention_sum(r_ctx, t))
        rw = np.sum(r_words, axis=-1)
        rw[0] = 0
        m = np.max(np.abs(rw))
        if m:
            rw = rw / m
        self.r_words = rw[1:]
        return R, self.attention[t]

    def explain_single_step(self, t):
        """E:438-535: truncated variant (no scan over earlier steps)."""
        if t > len(self.xt):
            raise NotImplementedError("index out of range of captions")
        w, H, E = self.w, self.H, self.E
        W_g = _gate_g_weight(w["lstm_Wi"], w["lstm_Wh"])
        r_h, r_ctx, r_s = self._head(t)
        r_c = r_h + r_s
        r_g = linear_lrp(r_c, self.it_act[t] * np.tanh(self.gt[t]), self.ct[t], np.identity(H))
        xh = np.hstack((self.xt[t - 1:t], self.ht[t - 1:t]))[0]
        r_xh = linear_lrp(r_g, xh, self.gt[t], W_g)
        R = self._image_tail(r_xh[E:2 * E], self._attention_sum(r_ctx, t))
        return R, self.attention[t]

    def explain_sentence(self):
        """E:183-189."""
        out = [self.explain(i + 1)[0] for i in range(len(self.caption) - 1)]
        return out, self.attention[1:-1]


class GridTDOracle(_Base):
    """ExplainImgCaptioningGridTDModel (E:995-1321) restated, quirks included:
    logits cached from h2 alone (E:1154) but the output rule is fed h2+c_hat
    (E:1212-1217); '+=' routing (E:1252-1254, E:1288, E:1300); r_words not
    normalised (E:1320)."""

    def forward(self, feat, caption, sos=2):
        """E:1092-1178."""
        w, H, E, L = self.w, self.H, self.E, self.L
        self.caption = list(caption)
        self._static_image_part(feat)
        self.proj = np.dot(self.Vfeat, w["W_va"])
        z32 = lambda *s: np.zeros(s, dtype="float32")
        for n in ("h1t", "c1t", "h2t", "c2t", "g1t", "i1t_act", "f1t_act", "g2t", "i2t_act", "f2t_act"):
            setattr(self, n, z32(1, H))
        self.context, self.st, self.context_hat = np.zeros((1, H)), np.zeros((1, H)), np.zeros((1, H))
        self.beta, self.attention = np.zeros((1, 1)), np.zeros((1, L))
        x1s, x2s, preds = [], [], []
        for i in range(len(caption)):
            h1p, c1p = self.h1t[-1].reshape(1, H), self.c1t[-1].reshape(1, H)
            h2p, c2p = self.h2t[-1].reshape(1, H), self.c2t[-1].reshape(1, H)
            tok = (sos if i == 0 else caption[i - 1]) - 1
            x1 = np.hstack((h2p, self.glob.reshape(1, E), w["embedding"][tok][None]))
            h1, c1, g1, i1, f1 = lstm_step(x1, h1p, c1p, w["td_Wi"], w["td_Wh"], w["td_b"], H)
            h_proj = np.dot(h1, w["W_ha"])
            att_pre = np.dot(np.tanh(self.proj + h_proj), w["W_a"])
            att = softmax(att_pre, axis=0)
            ctx = np.sum(att * self.Vfeat, axis=0).reshape(1, H)
            s = np.tanh(c1) * expit(np.dot(x1, w["W_x"]) + np.dot(h1p, w["W_h"]))
            z_s = np.dot(np.tanh(np.dot(s, w["W_s"]) + h_proj), w["W_a"])
            beta = softmax(np.concatenate((att_pre, z_s), axis=0), axis=0)[-1][0]
            c_hat = beta * s + (1 - beta) * ctx
            x2 = np.hstack((c_hat, h1.reshape(1, H)))
            h2, c2, g2, i2, f2 = lstm_step(x2, h2p, c2p, w["lang_Wi"], w["lang_Wh"], w["lang_b"], H)
            preds.append(np.dot(h2, w["output_W"]) + w["output_b"])        # h2 only (quirk)
            x1s.append(x1)
            x2s.append(x2)
            for n, v in (("h1t", h1), ("c1t", c1), ("g1t", g1), ("i1t_act", i1), ("f1t_act", f1),
                         ("h2t", h2), ("c2t", c2), ("g2t", g2), ("i2t_act", i2), ("f2t_act", f2),
                         ("context", ctx), ("st", s), ("context_hat", c_hat)):
                setattr(self, n, np.vstack((getattr(self, n), v.reshape(1, H))))
            self.beta = np.vstack((self.beta, beta.reshape(1, 1)))
            self.attention = np.vstack((self.attention, att.reshape(1, L)))
        self.x1t, self.x2t = np.vstack(x1s), np.vstack(x2s)
        self.caption_preds = np.vstack(preds)

    def explain(self, t):
        """E:1180-1321."""
        if t > len(self.x1t):
            raise NotImplementedError("index out of range of captions")
        w, H, E, L = self.w, self.H, self.E, self.L
        I = np.identity
        k = self.caption[t - 1] - 1
        seed = np.zeros((1, self.caption_preds.shape[1]))
        seed[0, k] = self.caption_preds[t - 1, k]
        xh1 = np.hstack((self.x1t[0:t], self.h1t[0:t]))
        xh2 = np.hstack((self.x2t[0:t], self.h2t[0:t]))
        Wg_td = _gate_g_weight(w["td_Wi"], w["td_Wh"])
        Wg_lang = _gate_g_weight(w["lang_Wi"], w["lang_Wh"])
        r_V = np.zeros((L, H), dtype="float32")
        r_glob = np.zeros(E)
        r_c1, r_c2 = np.zeros((t + 1, H)), np.zeros((t + 1, H))
        r_h1, r_h2 = np.zeros((t + 1, H)), np.zeros((t + 1, H))
        r_ctx, r_chat = np.zeros((t, H)), np.zeros((t, H))
        r_words = np.zeros((t, E))
        u = self.h2t[t] + self.context_hat[t]
        r_u = linear_lrp(seed, u, self.caption_preds[t - 1], w["output_W"])
        r_h2[t] = linear_lrp(r_u, self.h2t[t], u, I(H))
        r_chat[t - 1] = linear_lrp(r_u, self.context_hat[t], u, I(H))
        for i in range(t)[::-1]:
            # language LSTM
            r_c2[i + 1] += r_h2[i + 1]
            r_g2 = linear_lrp(r_c2[i + 1], self.i2t_act[i + 1] * np.tanh(self.g2t[i + 1]), self.c2t[i + 1], I(H))
            r_c2[i] = linear_lrp(r_c2[i + 1], self.f2t_act[i + 1] * self.c2t[i], self.c2t[i + 1], I(H))
            r_xh2 = linear_lrp(r_g2, xh2[i], self.g2t[i + 1], Wg_lang)
            r_h1[i + 1] += r_xh2[H:2 * H]
            r_h2[i] += r_xh2[2 * H:]
            r_chat[i] += r_xh2[:H]
            b = self.beta[i + 1][0]
            r_s = linear_lrp(r_chat[i], b * self.st[i + 1], self.context_hat[i + 1], I(H))
            r_ctx[i] = linear_lrp(r_chat[i], self.context[i + 1] * (1 - b), self.context_hat[i + 1], I(H))
            # top-down LSTM
            r_c1[i + 1] += r_s
            r_c1[i + 1] += r_h1[i + 1]
            r_g1 = linear_lrp(r_c1[i + 1], self.i1t_act[i + 1] * np.tanh(self.g1t[i + 1]), self.c1t[i + 1], I(H))
            r_c1[i] = linear_lrp(r_c1[i + 1], self.f1t_act[i + 1] * self.c1t[i], self.c1t[i + 1], I(H))
            r_xh1 = linear_lrp(r_g1, xh1[i], self.g1t[i + 1], Wg_td)
            r_h2[i] += r_xh1[:H]
            r_glob += r_xh1[H:H + E]
            r_words[i] = r_xh1[H + E:H + 2 * E]
            for k_ in range(L):
                r_V[k_] += linear_lrp(r_ctx[i], self.Vfeat[k_] * self.attention[i + 1][k_],
                                      self.context[i + 1], I(H))
            r_h1[i] += r_xh1[H + 2 * E:]
        R = self._image_tail(r_glob, r_V)
        self.r_words = np.sum(r_words, axis=-1)
        return R, self.attention[t]

    def explain_sentence(self):
        out = [self.explain(i + 1)[0] for i in range(len(self.caption) - 1)]
        return out, self.attention[1:-1]
