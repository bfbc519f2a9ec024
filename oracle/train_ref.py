"""TEST INFRASTRUCTURE — CPU oracle of the LRP-inference fine-tune step.

Restates, in torch float64 with autograd, what `TrainingLRPInferenceAdaptive.run` does per batch once
`lrp_weight` is known (train.py:573-581):

    losses = keras_model.train_on_batch(X + [lrp_weight], [y, y])

for the model built by `ImgCaptioningAdaptiveAttentionLRPInferenceModel.build` (models/model.py:1340-1374):

  * the truncated VGG16 encoder with every layer trainable                              (M:1322-1338)
  * `image_features` = TimeDistributed(Dense(H, relu)) -> Dropout                        (M:1346-1348)
  * `global_img_feature` = Dense(E, relu) on the L-mean of the raw features -> Dropout   (M:1343-1344, :1350-1352)
  * Embedding                                                                            (M:70-89)
  * LSTM inside ExternalAttentionRNNWrapperLocalAttentionV3.step                         (M:573-600, constants M:602-604)
  * Dropout -> TimeDistributed(Dense(V)) = logits; second head logits * lrp_weight       (M:1364-1368)
  * loss 0.5 * CE(y, logits[:, :-1]) + 0.5 * CE(y, (logits * lrp_weight)[:, :-1]), softmax cross-entropy with
    logits, Keras' mean over the (B, T-1) entries (an all-zero label row contributes 0)  (M:95-103, :1370-1373)
  * Adam(lr, clipvalue=c): gradients clipped element-wise to [-c, c], then Keras Adam
    (beta 0.9 / 0.999, epsilon K.epsilon() = 1e-7, lr_t = lr sqrt(1 - b2^t) / (1 - b1^t), p -= lr_t m / (sqrt(v) + eps)).

Dropout masks are explicit inputs (tensors of 0 or 1/(1-p)); `None` = inference-mode identity.  Keras draws one mask
per LSTM gate for the inputs and for the recurrent state (LSTMCell.call, implementation 1); the wrapper calls
`cell.call` from inside the `K.rnn` loop body (M:582), so the sampling op sits in the loop and the masks are given per
step here: `lstm_in` (T, 4, B, 2E), `lstm_rec` (T, 4, B, H), gate order i, f, c, o (repeat one mask T times for
time-constant dropout).  The sentinel gate reads the un-dropped input (M:584).

PARITY UNPINNED: the reference's training step lives in TensorFlow/Keras, which cannot run here and has no fixtures in
the reference's tests.  What pins this file: (1) its logits equal oracle/decoder_ref.AdaptiveOracle.forward — which is
pinned by outputs of the reference's own code (tests/golden) — on the same weights (tests/test_oracle_train.py);
(2) autograd gradients checked by central differences.  Only tests/, __graft_entry__.smoke() and bench.py's CPU leg
may import this file.
"""
import numpy as np
import torch
import torch.nn.functional as F

DT = torch.float64

GRIDTD_PARAMS = ("image_features_W", "image_features_b", "global_W", "global_b", "embedding", "td_Wi", "td_Wh", "td_b",
                 "lang_Wi", "lang_Wh", "lang_b", "W_va", "W_ha", "W_a", "W_x", "W_h", "W_s", "output_W", "output_b")

DECODER_PARAMS = ("image_features_W", "image_features_b", "global_W", "global_b", "embedding", "lstm_Wi", "lstm_Wh",
                  "lstm_b", "Wv", "Wg", "V", "Wx", "Wh", "Ws", "output_W", "output_b")


def param_names(cnn_cfg, kind="adaptive"):
    names = []
    for name, _, _, _ in cnn_cfg:
        names += [name + "_W", name + "_b"]
    return names + list(DECODER_PARAMS if kind == "adaptive" else GRIDTD_PARAMS)


def _t(a):
    return torch.as_tensor(np.ascontiguousarray(a)).to(DT)


def cnn_features(w, cnn_cfg, images_nhwc):
    """Truncated encoder, NHWC in -> (B, L, D) feature rows (Reshape((L, D)), M:1337)."""
    x = images_nhwc.permute(0, 3, 1, 2)
    for name, _, _, pool_after in cnn_cfg:
        x = F.relu(F.conv2d(x, w[name + "_W"].permute(3, 2, 0, 1), w[name + "_b"], padding=1))
        if pool_after:
            x = F.max_pool2d(x, 2, 2)
    x = x.permute(0, 2, 3, 1)
    return x.reshape(x.shape[0], -1, x.shape[-1])


def decoder_logits(w, feat, cap_in, masks=None):
    """feat (B, L, D), cap_in (B, T) embedding rows -> logits (B, T, V).  M:1343-1365 + M:573-600."""
    masks = masks or {}
    B, T = cap_in.shape
    H = w["lstm_Wh"].shape[0]
    m = lambda k: masks.get(k)
    Vf = F.relu(feat @ w["image_features_W"] + w["image_features_b"])
    if m("image_features") is not None:
        Vf = Vf * m("image_features")
    glob = F.relu(feat.mean(dim=1) @ w["global_W"] + w["global_b"])
    if m("global") is not None:
        glob = glob * m("global")
    emb = w["embedding"][cap_in]                                          # (B, T, E)
    proj = Vf @ w["Wv"]                                                   # get_constants, M:602-604
    h = torch.zeros(B, H, dtype=DT)
    c = torch.zeros(B, H, dtype=DT)
    outs = []
    for t in range(T):
        x = torch.cat([emb[:, t], glob], dim=1)                           # M:581
        zs = []
        for g in range(4):                                                # keras LSTMCell.call, implementation 1
            xi = x if m("lstm_in") is None else x * m("lstm_in")[t, g]
            hi = h if m("lstm_rec") is None else h * m("lstm_rec")[t, g]
            zs.append(xi @ w["lstm_Wi"][:, g * H:(g + 1) * H] + hi @ w["lstm_Wh"][:, g * H:(g + 1) * H]
                      + w["lstm_b"][g * H:(g + 1) * H])
        i_, f_, g_, o_ = torch.sigmoid(zs[0]), torch.sigmoid(zs[1]), torch.tanh(zs[2]), torch.sigmoid(zs[3])
        c_new = f_ * c + i_ * g_
        h_new = o_ * torch.tanh(c_new)
        s = torch.tanh(c_new) * torch.sigmoid(x @ w["Wx"] + h @ w["Wh"])              # M:584 (h = h_{t-1})
        hw = h_new @ w["Wg"]                                                           # M:589
        z_s = torch.tanh(s @ w["Ws"] + hw) @ w["V"]                                    # (B, 1)  M:586
        e = (torch.tanh(proj + hw[:, None, :]) @ w["V"])[..., 0]                       # (B, L)  M:590-592
        alpha = torch.softmax(e, dim=1)
        beta = torch.softmax(torch.cat([e, z_s], dim=1), dim=1)[:, -1:]                # M:593-595
        ctx = (alpha[..., None] * Vf).sum(dim=1)                                       # M:596
        c_hat = beta * s + (1 - beta) * ctx                                            # M:597
        outs.append(h_new + c_hat)                                                     # M:599
        h, c = h_new, c_new
    out = torch.stack(outs, dim=1)                                        # (B, T, H)
    if m("output") is not None:
        out = out * m("output")
    return out @ w["output_W"] + w["output_b"]


def decoder_logits_gridtd(w, feat, cap_in, masks=None):
    """Grid-TD (bottom-up / top-down) captioner in training mode: ImgCaptioningGridTDLRPInferenceModel.build (M:1275-1311)
    + ExternalBottomUpAttentionAdaptive.step (M:784-818) with its hand-written top-down cell (M:668-682).  Unlike the
    adaptive model there is a Dropout on the logits as well (M:1303-1304): mask 'logits' (B, T, V).  The language LSTM is a
    keras cell called inside the loop (M:812): per-step, per-gate masks 'lstm_in' (T, 4, B, 2H) / 'lstm_rec' (T, 4, B, H)."""
    masks = masks or {}
    B, T = cap_in.shape
    H = w["td_Wh"].shape[0]
    m = lambda k: masks.get(k)
    Vf = F.relu(feat @ w["image_features_W"] + w["image_features_b"])
    if m("image_features") is not None:
        Vf = Vf * m("image_features")
    glob = F.relu(feat.mean(dim=1) @ w["global_W"] + w["global_b"])
    if m("global") is not None:
        glob = glob * m("global")
    emb = w["embedding"][cap_in]
    proj = Vf @ w["W_va"]
    z = lambda: torch.zeros(B, H, dtype=DT)
    h1, c1, h2, c2 = z(), z(), z(), z()
    outs = []
    for t in range(T):
        x1 = torch.cat([h2, glob, emb[:, t]], dim=1)                                   # M:792
        zz = x1 @ w["td_Wi"] + h1 @ w["td_Wh"] + w["td_b"]                             # M:668-682
        i_, f_, g_, o_ = torch.sigmoid(zz[:, :H]), torch.sigmoid(zz[:, H:2 * H]), torch.tanh(zz[:, 2 * H:3 * H]), torch.sigmoid(zz[:, 3 * H:])
        c1n = f_ * c1 + i_ * g_
        h1n = o_ * torch.tanh(c1n)
        s = torch.tanh(c1n) * torch.sigmoid(x1 @ w["W_x"] + h1 @ w["W_h"])             # M:797 (h1 = previous state)
        hw = h1n @ w["W_ha"]
        z_s = torch.tanh(s @ w["W_s"] + hw) @ w["W_a"]                                 # M:799
        e = (torch.tanh(proj + hw[:, None, :]) @ w["W_a"])[..., 0]                     # M:800-801
        alpha = torch.softmax(e, dim=1)
        beta = torch.softmax(torch.cat([e, z_s], dim=1), dim=1)[:, -1:]                # M:804-806
        ctx = (alpha[..., None] * Vf).sum(dim=1)
        c_hat = beta * s + (1 - beta) * ctx                                            # M:808
        x2 = torch.cat([c_hat, h1n], dim=1)                                            # M:810
        zs = []
        for g in range(4):
            xi = x2 if m("lstm_in") is None else x2 * m("lstm_in")[t, g]
            hi = h2 if m("lstm_rec") is None else h2 * m("lstm_rec")[t, g]
            zs.append(xi @ w["lang_Wi"][:, g * H:(g + 1) * H] + hi @ w["lang_Wh"][:, g * H:(g + 1) * H] + w["lang_b"][g * H:(g + 1) * H])
        i2, f2, g2, o2 = torch.sigmoid(zs[0]), torch.sigmoid(zs[1]), torch.tanh(zs[2]), torch.sigmoid(zs[3])
        c2n = f2 * c2 + i2 * g2
        h2n = o2 * torch.tanh(c2n)
        outs.append(h2n + c_hat)                                                       # M:816
        h1, c1, h2, c2 = h1n, c1n, h2n, c2n
    out = torch.stack(outs, dim=1)
    if m("output") is not None:
        out = out * m("output")
    logits = out @ w["output_W"] + w["output_b"]
    if m("logits") is not None:
        logits = logits * m("logits")
    return logits


def two_head_loss(logits, lrp_weight, y_idx):
    """M:95-103 with loss_weights [0.5, 0.5] (M:1370-1373).  y_idx (B, T) class index, -1 = all-zero label row."""
    B, T, V = logits.shape
    y = torch.zeros(B, T, V, dtype=DT)
    valid = y_idx >= 0
    y[valid.nonzero(as_tuple=True) + (y_idx[valid],)] = 1.0
    def ce(z):
        return -(y[:, :-1] * torch.log_softmax(z[:, :-1], dim=-1)).sum(-1).mean()
    l1, l2 = ce(logits), ce(logits * lrp_weight)
    return 0.5 * l1 + 0.5 * l2, l1, l2


def loss_and_grads(weights, cnn_cfg, images, cap_in, y_idx, lrp_weight, masks=None, kind="adaptive"):
    """-> (total, l1, l2, {name: grad ndarray}) for every parameter of the training model."""
    w = {k: _t(v).requires_grad_(True) for k, v in weights.items() if k in set(param_names(cnn_cfg, kind))}
    mk = {k: _t(v) for k, v in (masks or {}).items() if v is not None}
    feat = cnn_features(w, cnn_cfg, _t(images))
    fwd = decoder_logits if kind == "adaptive" else decoder_logits_gridtd
    logits = fwd(w, feat, torch.as_tensor(np.asarray(cap_in)).long(), mk)
    total, l1, l2 = two_head_loss(logits, _t(lrp_weight), torch.as_tensor(np.asarray(y_idx)).long())
    total.backward()
    grads = {k: (v.grad.numpy() if v.grad is not None else np.zeros(tuple(v.shape))) for k, v in w.items()}
    return float(total.detach()), float(l1.detach()), float(l2.detach()), grads, logits.detach().numpy()


def adam_clipvalue_step(p, g, m, v, step, lr, clipvalue, b1=0.9, b2=0.999, eps=1e-7):
    """keras.optimizers.Adam.get_updates with clipvalue (keras 2.2.4 optimizers.py: clip, then moments, then update);
    `step` is the 1-based iteration count.  Arrays are updated out of place and returned."""
    g = np.clip(np.asarray(g, np.float64), -clipvalue, clipvalue) if clipvalue else np.asarray(g, np.float64)
    lr_t = lr * np.sqrt(1.0 - b2 ** step) / (1.0 - b1 ** step)
    m = b1 * m + (1.0 - b1) * g
    v = b2 * v + (1.0 - b2) * g * g
    return p - lr_t * m / (np.sqrt(v) + eps), m, v
