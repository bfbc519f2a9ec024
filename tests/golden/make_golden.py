#!/usr/bin/env python3
"""Generate golden vectors for the decoder half of the LRP hot path by running
the REFERENCE's own numpy code (models/explainers.py) in this container.

This is tooling, not a test and not product code.  It only runs where
/root/reference exists (the build container); the GPU box never runs it.
Nothing from the reference is copied: the reference modules are imported from
where they lie, driven with seeded synthetic weights, and only their numeric
inputs/outputs are written to tests/golden/*.npz.

How the import works: keras / tensorflow / skimage /
nltk / h5py / ... are absent here, so a sys.meta_path finder fabricates inert
stub modules for exactly those third-party packages.  The LRP arithmetic in
models/explainers.py (E:125-165, E:370-436, E:537-666, E:1092-1321) is pure
numpy/scipy and executes unmodified.

Usage:  python tests/golden/make_golden.py [--ref /root/reference]
"""
import argparse
import importlib.abc
import importlib.machinery
import os
import sys
import types

import numpy as np

STUB_TOPLEVEL = {"keras", "tensorflow", "skimage", "nltk", "keras_applications",
                 "h5py", "future", "bert_score", "cv2", "tf_keras"}
STUB_PREFIXES = tuple("pycocoevalcap." + s for s in
                      ("bleu", "cider", "meteor", "rouge", "spice", "tokenizer"))


class _DummyMeta(type):
    def __getattr__(cls, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _make_dummy(name)

    def __call__(cls, *a, **k):
        return type.__call__(cls)

    def __iter__(cls):
        return iter(())


def _make_dummy(name):
    def _init(self, *a, **k):
        pass

    def _getattr(self, n):
        if n.startswith("__") and n.endswith("__"):
            raise AttributeError(n)
        return _make_dummy(n)

    return _DummyMeta(str(name), (object,), {
        "__init__": _init, "__getattr__": _getattr,
        "__call__": lambda self, *a, **k: self,
        "__iter__": lambda self: iter(()),
    })


class _StubModule(types.ModuleType):
    __path__ = []

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        full = self.__name__ + "." + name
        if full in sys.modules:
            return sys.modules[full]
        if name == "epsilon":          # binds the default eps of E:157
            return lambda: 1e-7
        if name == "floatx":
            return lambda: "float32"
        if name == "image_data_format":
            return lambda: "channels_last"
        return _make_dummy(name)


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        top = fullname.split(".")[0]
        if top in STUB_TOPLEVEL or fullname.startswith(STUB_PREFIXES):
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        return _StubModule(spec.name)

    def exec_module(self, module):
        pass


def import_reference(ref_root):
    sys.meta_path.insert(0, _StubFinder())
    sys.path.insert(0, ref_root)
    import matplotlib
    matplotlib.use("Agg")
    import models.explainers as E  # noqa
    assert E.ExplainImgCaptioningAttentionModel._propagate_relevance_linear_lrp.__defaults__[1] == 1e-7
    return E


sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lrp_imagecaptioning_amd.synthetic import canned_next_word_scores, canned_score_table, decoder_case  # noqa: E402  (seeded generators shared with the tests)


def build_adaptive(E, w, feat, L, D, H, E_):
    o = object.__new__(E.ExplainImgCaptioningAdaptiveAttention)
    o.L, o.D, o._hidden_dim, o._embedding_dim = L, D, H, E_
    o._image_features_wieght, o._image_features_bias = w["image_features_W"], w["image_features_b"]
    o._global_img_feature_weight, o._global_img_feature_bias = w["global_W"], w["global_b"]
    o._lstm_weight_i, o._lstm_weight_h, o._lstm_bias = w["lstm_Wi"], w["lstm_Wh"], w["lstm_b"]
    o._Wv, o._Wg, o._V, o._Wx, o._Wh, o._Ws = w["Wv"], w["Wg"], w["V"], w["Wx"], w["Wh"], w["Ws"]
    o._output_weight, o._output_bias = w["output_W"], w["output_b"]
    o._image_model = types.SimpleNamespace(predict=lambda x: feat)
    emb = w["embedding"]
    o._embedding = types.SimpleNamespace(predict=lambda idx: emb[np.asarray(idx)][None])
    o._preprocessor = types.SimpleNamespace(SOS_TOKEN_LABEL_ENCODED=2, EOS_TOKEN_LABEL_ENCODED=1)
    return o


def build_gridtd(E, w, feat, L, D, H, E_):
    o = object.__new__(E.ExplainImgCaptioningGridTDModel)
    o.L, o.D, o._hidden_dim, o._embedding_dim = L, D, H, E_
    o._image_features_weight_bm, o._image_features_bias_bm = w["image_features_W"], w["image_features_b"]
    o._global_img_feature_weight_bm, o._global_img_feature_bias_bm = w["global_W"], w["global_b"]
    o._top_down_lstm_weight_i, o._top_down_lstm_weight_h = w["td_Wi"], w["td_Wh"]
    o._top_down_lstm_weight_bias = w["td_b"]
    o._language_lstm_weight_i, o._language_lstm_weight_h = w["lang_Wi"], w["lang_Wh"]
    o._language_lstm_bias = w["lang_b"]
    o._W_va, o._W_ha, o._W_a = w["W_va"], w["W_ha"], w["W_a"]
    o._W_x, o._W_h, o._W_s = w["W_x"], w["W_h"], w["W_s"]
    o._output_weight_bm, o._output_bias_bm = w["output_W"], w["output_b"]
    o._image_model = types.SimpleNamespace(predict=lambda x: feat)
    emb = w["embedding"]
    o._embedding_bm = types.SimpleNamespace(predict=lambda idx: emb[np.asarray(idx)][None])
    o._preprocessor = types.SimpleNamespace(SOS_TOKEN_LABEL_ENCODED=2, EOS_TOKEN_LABEL_ENCODED=1)
    return o


ADAPTIVE_STATE = ["ht", "ct", "gt", "it_act", "ft_act", "context", "attention", "st", "beta", "c_hat",
                  "xt", "caption_preds", "_image_features_before_act", "_average_img_feature",
                  "_global_img_feature_before_act", "_total_static_img_feature"]
GRIDTD_STATE = ["h1t", "c1t", "g1t", "i1t_act", "f1t_act", "h2t", "c2t", "g2t", "i2t_act", "f2t_act",
                "x1t", "x2t", "context", "st", "beta", "context_hat", "attention", "caption_preds",
                "_image_features_before_act_bm", "_average_img_feature_bm",
                "_global_image_feature_before_act_bm", "_image_features_proj_bm"]


def run_case(E, kind, seed, L, D, H, V, T, store_weights=True, tokens=None, single_word=False):
    E_ = H
    w, feat, cap = decoder_case(kind, seed, L, D, H, V, T)
    o = (build_adaptive if kind == "adaptive" else build_gridtd)(E, w, feat, L, D, H, E_)
    o._forward_beam_search((None, None), cap)
    out = {"kind": kind, "seed": seed, "dims": np.array([L, D, H, E_, V, T]),
           "feat": feat, "caption": np.array(cap, dtype=np.int64)}
    if store_weights:
        for k, v in w.items():
            out["w_" + k] = v
    for name in (ADAPTIVE_STATE if kind == "adaptive" else GRIDTD_STATE):
        out["state_" + name.lstrip("_")] = np.asarray(getattr(o, name))
    toks = tokens if tokens is not None else list(range(1, len(cap)))
    out["tokens"] = np.array(toks, dtype=np.int64)
    Rs, atts, rws = [], [], []
    for t in toks:
        R, att = o._explain_lstm_single_word_sequence(t)
        Rs.append(R.copy())
        atts.append(np.array(att, copy=True))
        rws.append(np.array(o.r_words, dtype=np.float64, copy=True))
    out["R_feat"] = np.stack(Rs)                      # (n_tok, 1, sqrtL, sqrtL, D) float32
    out["attention_t"] = np.stack(atts)
    for t, rw in zip(toks, rws):
        out["r_words_t%d" % t] = rw
    if single_word and kind == "adaptive":            # E:438-535 truncated variant
        Rs1 = []
        for t in toks:
            R, _ = o._explain_lstm_single_word(t)
            Rs1.append(R.copy())
        out["R_feat_single"] = np.stack(Rs1)
    # try the full caller too: _explain_sentence (E:183-189)
    if tokens is None:
        rel, att = o._explain_sentence()
        assert all(np.array_equal(a, b) for a, b in zip(rel, Rs))
        out["sentence_attention"] = np.asarray(att)
    return out


def run_gradient_case(E, kind, seed, L, D, H, V, T, store_weights=True, tokens=None):
    """The hand-written BPTT of the gradient baselines (E:780-832 adaptive, E:1452-1532 grid-TD) on the same
    seeded inputs: d(logit_k)/d(image features) per token and the per-word sums r_words."""
    E_ = H
    w, feat, cap = decoder_case(kind, seed, L, D, H, V, T)
    if kind == "adaptive":
        o = build_adaptive(E, w, feat, L, D, H, E_)
        o.__class__ = E.ExplainImgCaptioningAdaptiveAttentionGradient
    else:
        o = build_gridtd(E, w, feat, L, D, H, E_)
        o.__class__ = E.ExplainImgCaptioningGridTDGradient
    o._forward_beam_search((None, None), cap)
    out = {"kind": kind, "seed": seed, "dims": np.array([L, D, H, E_, V, T]),
           "feat": feat, "caption": np.array(cap, dtype=np.int64)}
    if store_weights:
        for k, v in w.items():
            out["w_" + k] = v
    toks = tokens if tokens is not None else list(range(1, len(cap)))
    out["tokens"] = np.array(toks, dtype=np.int64)
    gs = []
    for t in toks:
        g = o._lstm_decoder_backward(t)
        gs.append(np.array(g, copy=True))
        out["r_words_t%d" % t] = np.array(o.r_words, dtype=np.float64, copy=True)
    out["d_feat"] = np.stack(gs)                      # (n_tok, 1, sqrtL, sqrtL, D) float32
    for name in (("ot_act", "gt_act") if kind == "adaptive" else ("o1t_act", "o2t_act", "g1t_act", "g2t_act")):
        out["state_" + name] = np.asarray(getattr(o, name))
    if tokens is None:
        rel = o._explain_sentence()                   # E:834-839 / E:1534-1539
        assert all(np.array_equal(a, b) for a, b in zip(rel, gs))
    return out


def run_beam_case(E, seed, V, n_images, beam, max_len):
    """The reference's own caption search (`_beam_search`, E:51-120, with inference.py's BatchNLargest) on canned
    scores: `_keras_model.predict_on_batch` and `_preprocessor.preprocess_batch` are replaced by a seeded table lookup
    (lrp_imagecaptioning_amd.synthetic.canned_next_word_scores), everything else — log-soft-max, argpartition, the two
    bounded heaps, EOS handling, the final pick — is the reference's code.  EOS = 1 is one of V = 12 words, so
    hypotheses that produced EOS compete for beam slots at every step."""
    SOS, EOS = 2, 1
    table = canned_score_table(seed, V, n_images)

    def predict_on_batch(inputs):
        sentences = np.asarray(inputs[0])                 # (batch, len): [SOS, w..., EOS]; row i = image i
        out = np.zeros((len(sentences), 2, V), dtype=np.float32)
        for i, sent in enumerate(sentences):
            out[i, 0] = canned_next_word_scores(table, i, [int(w) for w in sent[1:-1]])
        return out                                        # E:74-75 keep [:, :-1][:, -1] = row 0

    o = object.__new__(E.ExplainImgCaptioningAttentionModel)
    o._preprocessor = types.SimpleNamespace(SOS_TOKEN_LABEL_ENCODED=SOS, EOS_TOKEN_LABEL_ENCODED=EOS,
                                            preprocess_batch=lambda sents: (np.asarray(sents), None))
    o._keras_model = types.SimpleNamespace(predict_on_batch=predict_on_batch)
    o._max_caption_length = max_len
    res = o._beam_search((None, np.zeros((n_images, 1))), beam_size=beam)
    assert len(res) == n_images
    out = {"seed": seed, "V": V, "n_images": n_images, "beam": beam, "max_len": max_len, "sos": SOS, "eos": EOS}
    for i, cap in enumerate(res):
        out["caption_%d" % i] = np.array(cap, dtype=np.int64)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.dirname(os.path.abspath(__file__)))
    ap.add_argument("--only", default="all", choices=["all", "lrp", "grad", "beam"])
    args = ap.parse_args()
    E = import_reference(args.ref)
    if args.only in ("all", "beam"):
        for seed, V, n_img, beam, max_len in [(0, 12, 6, 3, 8), (1, 12, 6, 3, 8), (2, 9, 5, 4, 6), (3, 30, 4, 2, 10)]:
            out = run_beam_case(E, seed, V, n_img, beam, max_len)
            path = os.path.join(args.out, "beam_s%d.npz" % seed)
            np.savez_compressed(path, **out)
            print("beam_s%d: %s" % (seed, [list(out["caption_%d" % i]) for i in range(n_img)]))
    if args.only == "beam":
        return
    if args.only in ("all", "grad"):
        gcases = [
            ("adaptive_grad_small_s0", "adaptive", 0, 16, 24, 32, 50, 6, dict()),
            ("adaptive_grad_small_s1", "adaptive", 1, 9, 32, 32, 40, 1, dict()),
            ("adaptive_grad_small_s2", "adaptive", 2, 16, 64, 64, 120, 9, dict()),
            ("gridtd_grad_small_s0", "gridtd", 0, 16, 24, 32, 50, 6, dict()),
            ("gridtd_grad_small_s1", "gridtd", 1, 9, 32, 32, 40, 1, dict()),
            ("gridtd_grad_small_s2", "gridtd", 2, 16, 64, 64, 120, 9, dict()),
            ("adaptive_grad_full_s0", "adaptive", 0, 196, 512, 512, 2000, 10, dict(store_weights=False, tokens=[1, 10])),
            ("gridtd_grad_full_s0", "gridtd", 0, 196, 512, 512, 2000, 10, dict(store_weights=False, tokens=[1, 10])),
        ]
        for name, kind, seed, L, D, H, V, T, kw in gcases:
            out = run_gradient_case(E, kind, seed, L, D, H, V, T, **kw)
            if name.endswith("full_s0"):
                for k in list(out):
                    if k.startswith("state_") and out[k].size > 20000:
                        del out[k]
                del out["feat"]
            path = os.path.join(args.out, name + ".npz")
            np.savez_compressed(path, **out)
            print("%-24s tokens=%s  sum|d|=%s  size=%.1f KB" % (
                name, list(out["tokens"]), np.round(np.abs(out["d_feat"]).reshape(len(out["d_feat"]), -1).sum(1), 5)[:3],
                os.path.getsize(path) / 1024))
    if args.only == "grad":
        return
    cases = [
        # name, kind, seed, L, D, H, V, T, kwargs
        ("adaptive_small_s0", "adaptive", 0, 16, 24, 32, 50, 6, dict(single_word=True)),
        ("adaptive_small_s1", "adaptive", 1, 9, 32, 32, 40, 1, dict(single_word=True)),   # T=1 edge case
        ("adaptive_small_s2", "adaptive", 2, 16, 64, 64, 120, 9, dict()),
        ("gridtd_small_s0", "gridtd", 0, 16, 24, 32, 50, 6, dict()),
        ("gridtd_small_s1", "gridtd", 1, 9, 32, 32, 40, 1, dict()),
        ("gridtd_small_s2", "gridtd", 2, 16, 64, 64, 120, 9, dict()),
        # full-size (L=196, D=H=E=512): weights are NOT stored (rebuilt from the
        # seed by the same generator); two tokens only to keep the file small
        ("adaptive_full_s0", "adaptive", 0, 196, 512, 512, 2000, 10,
         dict(store_weights=False, tokens=[1, 10])),
        ("gridtd_full_s0", "gridtd", 0, 196, 512, 512, 2000, 10,
         dict(store_weights=False, tokens=[1, 10])),
    ]
    for name, kind, seed, L, D, H, V, T, kw in cases:
        out = run_case(E, kind, seed, L, D, H, V, T, **kw)
        if name.endswith("full_s0"):
            # keep the file small: drop the big (T,V)/(L,H) state arrays, keep scalars+R
            for k in list(out):
                if k.startswith("state_") and out[k].size > 20000:
                    del out[k]
            del out["feat"]
        path = os.path.join(args.out, name + ".npz")
        np.savez_compressed(path, **out)
        R = out["R_feat"]
        print("%-22s tokens=%s  sum(R)=%s  size=%.1f KB" % (
            name, list(out["tokens"]), np.round(R.reshape(len(R), -1).sum(1), 6)[:3],
            os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
