"""Engine classes with the reference's explainer protocol (models/explainers.py, cited E:).

    ExplainImgCaptioningAttentionModel      E:22-256   (base: CNN analyzer, _explain_CNN, _explain_sentence)
    ExplainImgCaptioningAdaptiveAttention   E:260-666
    ExplainImgCaptioningGridTDModel         E:995-1321

Same method names, argument meaning, return shapes/dtypes and error behaviour, so
that `explain_image.Explainer`-style harnesses (explain_image.py:4-183) can drive
them unchanged.  What differs is construction: there is no Keras model to pull
weights from, so `model` is a `CaptionModelSpec` (geometry + a dict of arrays in
the Keras layouts) and `weight_path` may name an `.npz` bundle of those arrays
(hdf5 is not available).  All arithmetic runs in liblrp_hip.so through LRPEngine;
these classes hold no math of their own.
"""
import numpy as np
import torch

from .engine import LRPEngine
from .synthetic import VGG16_CFG

EPS = 0.01      # E:18  (CNN analyzer epsilon; only Dense layers use it, none in the truncated VGG16)
ALPHA = 1       # E:19
BETA = 0        # E:20


class CaptionPreprocessorStub(object):
    """The two attributes of models/preprocessors.py:CaptionPreprocessorAttention the hot path reads."""

    def __init__(self, sos=2, eos=1, word_of=None):
        self.SOS_TOKEN_LABEL_ENCODED = sos
        self.EOS_TOKEN_LABEL_ENCODED = eos
        self._word_of = word_of or {}

    def decode_captions_from_list1d(self, ids):
        return " ".join(self._word_of.get(int(i), "<%d>" % int(i)) for i in ids if int(i) != self.EOS_TOKEN_LABEL_ENCODED)


class DatasetProviderStub(object):
    def __init__(self, caption_preprocessor=None, image_preprocessor=None):
        self.caption_preprocessor = caption_preprocessor or CaptionPreprocessorStub()
        self.image_preprocessor = image_preprocessor


class CaptionModelSpec(object):
    """What the reference reads off `model` (E:26-39): encoder kind, dims, and the weights."""

    def __init__(self, weights, img_encoder="vgg16", hidden_dim=512, embedding_dim=512, L=196, D=512,
                 vocab_size=None, cnn_cfg=None, img_hw=(224, 224), resnet=None):
        """img_encoder 'vgg16' (config.py:36-40: block5_conv3, 196 x 512), 'vgg19' (config.py:37: block5_conv4) or
        'resnet101' (config.py:41-45: conv5_block3_out, 49 x 2048; `resnet` = dict(stem, stacks), default ResNet-101).
        cnn_cfg: the conv list of a VGG-style encoder (default: the full-size one of `img_encoder`)."""
        if img_encoder not in ("vgg16", "vgg19", "resnet101"):
            raise NotImplementedError("the img_encode is not valid, [vgg16, vgg19, resnet101]")     # explain_image.py:25-26
        if cnn_cfg is None:
            from .synthetic import VGG19_CFG
            cnn_cfg = VGG19_CFG if img_encoder == "vgg19" else VGG16_CFG
        if img_encoder == "resnet101" and resnet is None:
            from .synthetic import RESNET101_STACKS
            resnet = {"stem": 64, "stacks": RESNET101_STACKS}
        self.resnet = resnet if img_encoder == "resnet101" else None
        self.weights = dict(weights)
        self.img_encoder = img_encoder
        self._hidden_dim, self._embedding_dim = hidden_dim, embedding_dim
        self.L, self.D = L, D
        self.vocab_size = vocab_size if vocab_size is not None else int(self.weights["output_W"].shape[1])
        self.cnn_cfg = list(cnn_cfg)
        self.img_hw = tuple(img_hw)


class ExplainImgCaptioningAttentionModel(object):
    _decoder_kind = None

    def __init__(self, model, weight_path=None, dataset_provider=None, max_caption_length=20, max_images=1,
                 device=None):
        if weight_path:
            with np.load(weight_path) as z:               # stands in for keras load_weights (E:27)
                model.weights.update({k: z[k] for k in z.files})
        self._model = model
        self._img_encoder = model.img_encoder
        self._dataset_provider = dataset_provider or DatasetProviderStub()
        self._preprocessor = self._dataset_provider.caption_preprocessor
        self._max_caption_length = max_caption_length
        self._hidden_dim, self._embedding_dim = model._hidden_dim, model._embedding_dim
        self.L, self.D = model.L, model.D
        self._weight_path = (weight_path or "").strip("npz")
        Tm = max_caption_length + 1
        self._engine = LRPEngine(decoder=self._decoder_kind, cnn_cfg=model.cnn_cfg, img_hw=model.img_hw, L=model.L,
                                 D=model.D, H=model._hidden_dim, E=model._embedding_dim, V=model.vocab_size,
                                 max_images=max_images, max_tokens=max_images * Tm, max_caption_len=Tm,
                                 sos_id=self._preprocessor.SOS_TOKEN_LABEL_ENCODED,
                                 eos_id=self._preprocessor.EOS_TOKEN_LABEL_ENCODED, device=device,
                                 resnet=getattr(model, "resnet", None))
        self._engine.set_weights(model.weights)
        self._CNN_explainer = _EngineAnalyzer(self._engine)      # LRPSequentialPresetA(image_model, EPS, 'replace'), E:32
        self._state_cache = {}
        self.caption = None
        self.r_words = None

    # -------------------------------------------------------------- cached forward state as attributes
    _STATE_ATTRS = ()

    def __getattr__(self, name):
        if name.startswith("__") or name not in type(self)._STATE_ATTRS:
            raise AttributeError(name)
        if self.caption is None:
            raise AttributeError("%s is only available after _forward_beam_search" % name)
        if name not in self._state_cache:
            self._state_cache[name] = self._fetch_state(name)
        return self._state_cache[name]

    def _fetch_state(self, name):
        raise NotImplementedError()

    # -------------------------------------------------------------- protocol
    def _forward_beam_search(self, X, beam_search_captions):
        """E:370-436 / E:1092-1178: cache everything LRP needs for a given caption.
        X = (sequence_input, img_input); img_input (1,H,W,3) BGR mean-subtracted."""
        _, img_input = X
        self.caption = [int(c) for c in beam_search_captions]
        self._state_cache = {}
        self._img_input = np.asarray(img_input, dtype=np.float32)
        self._engine.encode_images(self._img_input[:1])
        self._engine.decoder_forward([self.caption])

    def _check_t(self, t):
        if self.caption is None:
            raise RuntimeError("_forward_beam_search must run first")
        if t > len(self.caption) or t < 1:
            raise NotImplementedError("index out of range of captions")          # E:538-539

    def _explain_lstm_single_word_sequence(self, t=0):
        """E:537-666 / E:1180-1321 -> (R (1,sqrtL,sqrtL,D) float32, attention_t (L,)); sets self.r_words."""
        self._check_t(t)
        R, att, rw = self._engine.decoder_explain([0], [t], variant="sequence")
        g = int(np.sqrt(self.L))
        self.r_words = self._r_words_from(rw[0].cpu().numpy(), t)
        return R.cpu().numpy().reshape(1, g, g, self.D), att[0].cpu().numpy()

    def _explain_lstm_single_word(self, t=0):
        """E:438-535 (one-step variant)."""
        self._check_t(t)
        R, att, _ = self._engine.decoder_explain([0], [t], variant="single_step", want_r_words=False)
        g = int(np.sqrt(self.L))
        return R.cpu().numpy().reshape(1, g, g, self.D), att[0].cpu().numpy()

    def _r_words_from(self, row, t):
        raise NotImplementedError()

    _batched_cnn = True      # _explain_CNN takes a stack of relevance rows (the harness then walks all words at once)

    def _explain_CNN(self, X, relevance_value, as_tensor=False):
        """E:179-181: `self._CNN_explainer.analyze([X, R])`.  The reference re-runs the whole
        encoder forward on every call (AB:511); when X is the image `_forward_beam_search`
        already encoded, the cached relevance gates are reused instead.  as_tensor=True keeps
        the (n,H,W,3) result on the device (for `engine.heatmap_render`)."""
        X = np.asarray(X, dtype=np.float32)
        R = np.asarray(relevance_value, dtype=np.float32)
        cached = getattr(self, "_img_input", None)
        if (self.caption is not None and cached is not None and X.shape[0] == 1 and X.shape == cached[:1].shape
                and np.array_equal(X, cached[:1])):
            n = R.shape[0]
            out = self._engine.cnn_explain([0] * n, R.reshape(n, self.L, self.D))
            return out if as_tensor else out.cpu().numpy()
        out = self._CNN_explainer.analyze([X, R])        # another image: the caches now belong to it
        self.caption = None
        self._state_cache = {}
        return torch.as_tensor(out).to(self._engine.device) if as_tensor else out

    def _explain_sentence(self):
        """E:183-189, but all tokens in ONE batched launch chain."""
        n = len(self.caption) - 1
        ts = list(range(1, n + 1))
        R, att, rw = self._engine.decoder_explain([0] * n, ts, variant="sequence")
        g = int(np.sqrt(self.L))
        Rn = R.cpu().numpy()
        if n:
            self.r_words = self._r_words_from(rw[n - 1].cpu().numpy(), n)
        rel = [Rn[i].reshape(1, g, g, self.D) for i in range(n)]
        return rel, self.attention[1:-1]

    # -------------------------------------------------------------- batched entry points (new)
    def explain_batch(self, images, captions, return_R_feat=False):
        """images (B,H,W,3), captions: list of id lists ending in EOS.  Returns
        (R_img (sum_b T_b, H, W, 3) torch tensor on the GPU, index list [(b, t)], attention, r_words)."""
        self._engine.encode_images(images)
        self._engine.decoder_forward(captions)
        pairs = [(b, t) for b, c in enumerate(captions) for t in range(1, len(c))]
        out, R, att, rw = self._engine.explain_tokens([p[0] for p in pairs], [p[1] for p in pairs],
                                                      want_R_feat=return_R_feat, want_attention=True, want_r_words=True)
        return out, pairs, att, rw, R

    def _beam_search(self, X, beam_size=3):
        """Caption generation (E:51-120; the batched form of inference.py:178-253).  Upstream of the LRP path (captions
        are an input to it); provided so harnesses run end to end.
          * bookkeeping = `beam.search`: the reference's two bounded heaps, EOS handling and final pick, pinned against
            the reference's own `_beam_search` on canned scores (tests/test_beam.py);
          * scores: the hypotheses' decoder state stays on the device (lrp_decoder_gen_begin / _gen_step: one decoder
            step per search step for ALL beams of ALL images that fit the handle — max_images >= images x beam_size —
            where the reference re-runs the whole captioner on every partial caption, E:71), and so does the ranking
            (lrp_op_log_softmax_topk): per step only beam_size (id, log p) pairs per hypothesis come to the host.
        Returns, per image, up to beam_size captions, element [0] = the reference's answer (one image: that list)."""
        from . import beam
        _, imgs_input = X
        imgs_input = np.asarray(imgs_input, dtype=np.float32)
        EOS = self._preprocessor.EOS_TOKEN_LABEL_ENCODED
        eng = self._engine
        k = beam_size
        if k > eng.max_images:
            return self._beam_search_replay(X, beam_size)          # not enough feature slots for one row per beam
        results = []
        group = max(1, eng.max_images // k)      # images searched together: one decoder step serves all their beams
        for lo in range(0, len(imgs_input), group):
            imgs = imgs_input[lo:lo + group]
            G = len(imgs)
            eng.encode_images(imgs)
            feat = eng.get_features()[:G]
            eng.set_features(feat.repeat_interleave(k, dim=0).contiguous())    # one row (feature slot) per (image, beam)
            eng.gen_begin(G * k)

            def step(s, parent, word):
                logits = eng.gen_step(s, parent, word)                          # (G * k, V) float64, stays on the device
                ids, logp = eng.log_softmax_topk(logits, k)
                return ids.cpu().numpy(), logp.cpu().numpy()
            results += beam.search(step, G, k, self._max_caption_length, EOS)
        self.caption = None
        self._state_cache = {}
        return results[0] if len(results) == 1 else results

    def _beam_search_replay(self, X, beam_size=3):
        """The same search with the scores taken from the teacher-forced decoder replay (every step re-plays the live
        hypotheses from scratch, like the reference's predict_on_batch loop) and ranked on the host: the cross-check of
        `_beam_search`'s incremental device state and device-side ranking."""
        from . import beam
        _, imgs_input = X
        imgs_input = np.asarray(imgs_input, dtype=np.float32)
        EOS = self._preprocessor.EOS_TOKEN_LABEL_ENCODED
        eng = self._engine
        k = beam_size
        results = []
        for img in imgs_input:
            eng.encode_images(img[None])
            hist = {}

            def step(s, parent, word):
                nonlocal hist
                hist = {r: [] for r in range(k)} if s == 0 else {r: hist[parent[r]] + [int(word[r])] for r in range(k)}
                rows = []
                for r in range(k):
                    eng.decoder_forward([hist[r] + [EOS]])
                    rows.append(eng.read_state("caption_preds")[0, s].cpu().numpy())
                return beam.topk_log_softmax(np.stack(rows), k)
            results += beam.search(step, 1, k, self._max_caption_length, EOS)
        self.caption = None
        self._state_cache = {}
        return results[0] if len(results) == 1 else results


class _EngineAnalyzer(object):
    """`analyze([X, R])` of LRPSequentialPresetA (AB:478-520) on an engine's encoder."""

    def __init__(self, engine):
        self._engine = engine

    def analyze(self, X):
        X = list(X) if isinstance(X, (list, tuple)) else [X]
        if len(X) != 2:
            raise ValueError("neuron_selection_mode 'replace' expects [X, R]")
        img, R = np.asarray(X[0], dtype=np.float32), np.asarray(X[1], dtype=np.float32)
        n = img.shape[0]
        if R.shape[0] != n:
            raise ValueError("X and R must have the same batch size")
        e = self._engine
        out = np.empty_like(img)
        for lo in range(0, n, e.max_images):
            hi = min(n, lo + e.max_images, lo + e.max_tokens)
            e.encode_images(img[lo:hi])
            out[lo:hi] = e.cnn_explain(list(range(hi - lo)), R[lo:hi].reshape(hi - lo, e.L, e.D)).cpu().numpy()
        return out


class ExplainImgCaptioningAdaptiveAttention(ExplainImgCaptioningAttentionModel):
    """E:260-666."""
    _decoder_kind = "adaptive"
    _STATE_ATTRS = ("ht", "ct", "gt", "it_act", "ft_act", "context", "attention", "st", "beta", "c_hat", "xt",
                    "caption_preds", "_image_features_before_act", "_average_img_feature",
                    "_global_img_feature_before_act", "_total_static_img_feature", "_img_feature_input")
    _ENGINE_NAME = {"_image_features_before_act": "image_features_before_act",
                    "_average_img_feature": "average_img_feature",
                    "_global_img_feature_before_act": "global_img_feature_before_act",
                    "_total_static_img_feature": "total_static_img_feature"}

    def _fetch_state(self, name):
        n = len(self.caption)
        if name == "_img_feature_input":
            return self._engine.get_features()[0].cpu().numpy()
        a = self._engine.read_state(self._ENGINE_NAME.get(name, name))[0].cpu().numpy()
        if name in ("xt", "caption_preds"):
            return a[:n]
        if name.startswith("_"):
            return a[0] if a.shape[0] == 1 else a
        return a[:n + 1]                              # row 0 = zero init (E:389-398)

    def _r_words_from(self, row, t):
        return row[:t - 1]                            # normalised, first dropped (E:660-665)


class ExplainImgCaptioningGridTDModel(ExplainImgCaptioningAttentionModel):
    """E:995-1321."""
    _decoder_kind = "gridtd"
    _STATE_ATTRS = ("h1t", "c1t", "g1t", "i1t_act", "f1t_act", "h2t", "c2t", "g2t", "i2t_act", "f2t_act", "x1t", "x2t",
                    "context", "st", "beta", "context_hat", "attention", "caption_preds")

    def _fetch_state(self, name):
        n = len(self.caption)
        a = self._engine.read_state(name)[0].cpu().numpy()
        if name in ("x1t", "x2t", "caption_preds"):
            return a[:n]
        return a[:n + 1]

    def _r_words_from(self, row, t):
        return row[:t]                                # un-normalised (E:1320)

    def _explain_lstm_single_word(self, t=0):
        raise NotImplementedError()                   # E:167-172: the grid-TD class does not override it


# ---------------------------------------------------------------------------------------------------------------
# Gradient baselines: the reference's six comparison engines on the same cached forward.
# Decoder half = its hand-written BPTT `_lstm_decoder_backward` (lrp_decoder_gradient); CNN half = iNNvestigate's
# Gradient / InputTimesGradient / GuidedBackprop with neuron_selection_mode="replace" (lrp_cnn_walk).
# ---------------------------------------------------------------------------------------------------------------
class _GradientMixin(object):
    _walk = "gradient"

    def __init__(self, *args, **kwargs):
        super(_GradientMixin, self).__init__(*args, **kwargs)
        # A gradient through ReLUs / max-pools switches whole paths where a pre-activation is zero to the forward's own
        # accuracy (LRP does not: such a unit carries ~ 0 relevance).  The comparison baselines therefore run the engine
        # with the forward that takes the fewest of those decisions differently from float64 — the exact fp32 MFMA
        # (their walks are exact fp32 in every mode): VGG16-size gradient maps 7e-7 from float64 instead of ~1e-3
        # (tests/test_gpu_gradient.py).  `self._engine.set_precision("bf16x3")` restores the faster default forward.
        self._engine.set_precision("fp32")

    def _lstm_decoder_backward(self, t):
        """E:780-832 / E:1452-1532 -> d_img_feature (1, sqrtL, sqrtL, D) float32; sets self.r_words (t,)."""
        self._check_t(t)
        d, rw = self._engine.decoder_gradient([0], [t])
        g = int(np.sqrt(self.L))
        self.r_words = rw[0, :t].cpu().numpy()
        return d.cpu().numpy().reshape(1, g, g, self.D)

    def _explain_sentence(self):
        """E:834-839 / E:1534-1539: a list of relevances only (no attention), all words in one batched call."""
        n = len(self.caption) - 1
        if n < 1:
            return []
        d, rw = self._engine.decoder_gradient([0] * n, list(range(1, n + 1)))
        g = int(np.sqrt(self.L))
        self.r_words = rw[n - 1, :n].cpu().numpy()
        dn = d.cpu().numpy()
        return [dn[i].reshape(1, g, g, self.D) for i in range(n)]

    def _explain_CNN(self, X, relevance_value, as_tensor=False):
        """`self._CNN_explainer.analyze([X, relevance])` with the class's analyzer (E:672 / :884 / :928)."""
        X = np.asarray(X, dtype=np.float32)
        R = np.asarray(relevance_value, dtype=np.float32)
        cached = getattr(self, "_img_input", None)
        if not (self.caption is not None and cached is not None and X.shape == cached[:1].shape and np.array_equal(X, cached[:1])):
            self._engine.encode_images(X[:1])              # another image: the caches now belong to it
            self._img_input = X[:1].copy()
            self.caption = None
            self._state_cache = {}
        n = R.shape[0]
        out = self._engine.cnn_walk([0] * n, R.reshape(n, self.L, self.D), self._walk)
        return out if as_tensor else out.cpu().numpy()

    def _explain_lstm_single_word_sequence(self, t=0):
        raise NotImplementedError("the gradient engines explain through _lstm_decoder_backward")   # E:174-177 base stub

    def _explain_lstm_single_word(self, t=0):
        raise NotImplementedError("the gradient engines explain through _lstm_decoder_backward")


class _GuidedGradcamMixin(_GradientMixin):
    _walk = "guided_backprop"
    _batched_cnn = False     # the Grad-CAM factor is per word (E:930-937)

    def grad_cam(self, img_feature, grads):
        from .postprocess import grad_cam
        return grad_cam(img_feature, grads, self.L, self.D, upscale=self._model.img_hw[0] // int(np.sqrt(self.L)))

    def _explain_CNN(self, X, relevance_value):
        """E:930-937 / E:1634-1641: guided backprop of the image model, gated by the Grad-CAM map of the features."""
        R = np.asarray(relevance_value, dtype=np.float32)
        gb = _GradientMixin._explain_CNN(self, X, R)
        feat = self._engine.get_features()[0].cpu().numpy()
        cam = self.grad_cam(feat, R[0])
        return (gb[0] * cam[..., np.newaxis])[np.newaxis, :]


class ExplainImgCaptioningAdaptiveAttentionGradient(_GradientMixin, ExplainImgCaptioningAdaptiveAttention):
    """E:667-879."""
    _STATE_ATTRS = ExplainImgCaptioningAdaptiveAttention._STATE_ATTRS + ("ot_act",)


class ExplainImgCaptioningAdaptiveAttentionInputTimesGradient(ExplainImgCaptioningAdaptiveAttentionGradient):
    """E:880-924."""
    _walk = "input_x_gradient"


class ExplainImgCaptioningAdaptiveAttentionGuidedGradcam(_GuidedGradcamMixin, ExplainImgCaptioningAdaptiveAttention):
    """E:925-993."""
    _STATE_ATTRS = ExplainImgCaptioningAdaptiveAttention._STATE_ATTRS + ("ot_act",)


class ExplainImgCaptioningGridTDGradient(_GradientMixin, ExplainImgCaptioningGridTDModel):
    """E:1322-1583."""
    _STATE_ATTRS = ExplainImgCaptioningGridTDModel._STATE_ATTRS + ("o1t_act", "o2t_act")


class ExplainImgCaptioningGridTDGradientTimesInput(ExplainImgCaptioningGridTDGradient):
    """E:1584-1628."""
    _walk = "input_x_gradient"


class ExplainImgCaptioningGridTDGuidedGradcam(_GuidedGradcamMixin, ExplainImgCaptioningGridTDModel):
    """E:1629-1700 (its guided_backproe helper, E:1655-1657, is `_explain_CNN` without the Grad-CAM factor)."""
    _STATE_ATTRS = ExplainImgCaptioningGridTDModel._STATE_ATTRS + ("o1t_act", "o2t_act")

    def guided_backproe(self, img_input, grads):
        return _GradientMixin._explain_CNN(self, img_input, grads)
