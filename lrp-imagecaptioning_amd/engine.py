"""LRPEngine — thin Python owner of one `lrp_handle` (one per GPU / stream).

PyTorch is plumbing only: it owns the device tensors that cross the C ABI
(images, relevance maps) and the current stream.  All arithmetic of the hot
path runs in liblrp_hip.so.
"""
import ctypes as C

import numpy as np
import torch

from . import _capi
from .synthetic import VGG16_CFG


def _i32(a):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.int32))
    return a, a.ctypes.data_as(C.POINTER(C.c_int32))


class LRPEngine(object):
    """Geometry + weights + caches for one captioning model on one GPU."""

    def __init__(self, decoder="adaptive", cnn_cfg=VGG16_CFG, img_hw=(224, 224), L=196, D=512, H=512, E=512,
                 V=10000, max_images=32, max_tokens=320, max_caption_len=21, sos_id=2, eos_id=1, device=None,
                 resnet=None):
        """resnet: None (VGG-style `cnn_cfg`) or dict(stem=64, stacks=((64,3),(128,4),(256,23),(512,3))) for the
        ResNet-v1 bottleneck encoder of BASELINE config 4 (then cnn_cfg is ignored)."""
        if decoder not in ("adaptive", "gridtd"):
            raise NotImplementedError("decoder must be 'adaptive' or 'gridtd'")
        self._lib = _capi.load()                      # raises if the HIP library is missing
        if not torch.cuda.is_available():
            raise RuntimeError("LRPEngine needs a ROCm GPU (no CPU fallback for the LRP hot path)")
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        cfg = _capi.LrpConfig()
        cfg.abi_version = _capi.LRP_ABI_VERSION
        cfg.device = self.device.index
        cfg.decoder = _capi.LRP_DEC_ADAPTIVE if decoder == "adaptive" else _capi.LRP_DEC_GRIDTD
        cfg.img_h, cfg.img_w = img_hw
        if resnet is None:
            cfg.encoder = _capi.LRP_ENC_VGG
            cfg.n_conv = len(cnn_cfg)
            for i, (name, cin, cout, pool) in enumerate(cnn_cfg):
                cfg.conv_cin[i], cfg.conv_cout[i], cfg.conv_pool_after[i] = cin, cout, int(bool(pool))
                cfg.conv_name[i].value = name.encode()
        else:
            cfg.encoder = _capi.LRP_ENC_RESNET
            cfg.resnet_stem = int(resnet.get("stem", 64))
            stacks = list(resnet["stacks"])
            cfg.resnet_n_stacks = len(stacks)
            for i, (f, nb) in enumerate(stacks):
                cfg.resnet_filters[i], cfg.resnet_blocks[i] = int(f), int(nb)
        cfg.L, cfg.D, cfg.H, cfg.E, cfg.V = L, D, H, E, V
        cfg.max_images, cfg.max_tokens, cfg.max_caption_len = max_images, max_tokens, max_caption_len
        cfg.sos_id, cfg.eos_id = sos_id, eos_id
        self.cfg = cfg
        self.decoder = decoder
        self.cnn_cfg = list(cnn_cfg)
        self.img_hw = tuple(img_hw)
        self.L, self.D, self.H, self.E, self.V = L, D, H, E, V
        self.max_images, self.max_tokens, self.Tm = max_images, max_tokens, max_caption_len
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            _capi.check(self._lib.lrp_create(C.byref(cfg), C.byref(self._h)))
        self.captions = None
        self.n_images = 0
        self.precision = "bf16x3"                       # library default for the reverse walk (set_precision)

    # ------------------------------------------------------------------ lifetime
    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.lrp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def workspace_bytes(self):
        return int(self._lib.lrp_workspace_bytes(self._h))

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _dev(self, x, dtype=torch.float32):
        if isinstance(x, torch.Tensor):
            return x.to(device=self.device, dtype=dtype).contiguous()
        return torch.as_tensor(np.ascontiguousarray(x), dtype=dtype).to(self.device)

    # ------------------------------------------------------------------ weights
    def set_weights(self, weights):
        """weights: dict name -> float32 array in the Keras layout (conv HWIO, dense (in,out))."""
        for name, arr in weights.items():
            if isinstance(arr, torch.Tensor):
                arr = arr.detach().cpu().numpy()
            a = np.ascontiguousarray(arr, dtype=np.float32)
            shape = (C.c_int64 * a.ndim)(*a.shape)
            _capi.check(self._lib.lrp_set_weight(self._h, name.encode(), a.ctypes.data_as(C.c_void_p), a.ndim, shape))

    def set_weights_from_device(self, weights):
        """Same, from CUDA tensors (e.g. after an RCCL broadcast of the frozen bundle)."""
        for name, t in weights.items():
            t = t.to(device=self.device, dtype=torch.float32).contiguous()
            shape = (C.c_int64 * t.dim())(*t.shape)
            _capi.check(self._lib.lrp_set_weight_dev(self._h, name.encode(), C.c_void_p(t.data_ptr()), t.dim(), shape,
                                                     self._stream()))

    # ------------------------------------------------------------------ encoder
    def encode_images(self, images):
        """images (B,H,W,3) float32 BGR mean-subtracted -> caches; returns nothing."""
        x = self._dev(images)
        if x.dim() != 4 or tuple(x.shape[1:]) != (self.img_hw[0], self.img_hw[1], 3):
            raise ValueError("images must be (B,%d,%d,3)" % self.img_hw)
        _capi.check(self._lib.lrp_encode_images(self._h, C.c_void_p(x.data_ptr()), x.shape[0], self._stream()))
        self.n_images = int(x.shape[0])
        self.captions = None

    def set_features(self, feat):
        f = self._dev(feat).reshape(-1, self.L, self.D)
        _capi.check(self._lib.lrp_set_features(self._h, C.c_void_p(f.data_ptr()), f.shape[0], self._stream()))
        self.n_images = int(f.shape[0])
        self.captions = None

    def get_features(self):
        out = torch.empty((self.n_images, self.L, self.D), dtype=torch.float32, device=self.device)
        _capi.check(self._lib.lrp_get_features(self._h, C.c_void_p(out.data_ptr()), self.n_images, self._stream()))
        return out

    # ------------------------------------------------------------------ decoder
    def decoder_forward(self, captions):
        """captions: list (per image) of tokenizer-id lists ending in EOS."""
        B = len(captions)
        caps = np.full((B, self.Tm), int(self.cfg.eos_id), dtype=np.int32)
        lens = np.zeros((B,), dtype=np.int32)
        for b, c in enumerate(captions):
            if len(c) > self.Tm:
                raise ValueError("caption %d longer than max_caption_len=%d" % (b, self.Tm))
            caps[b, :len(c)] = c
            lens[b] = len(c)
        caps, pc = _i32(caps)
        lens, pl = _i32(lens)
        _capi.check(self._lib.lrp_decoder_forward(self._h, pc, pl, B, self._stream()))
        self.captions = [list(map(int, c)) for c in captions]

    _STATE_SHAPES = {
        "ht": ("S", "H", torch.float32), "ct": ("S", "H", torch.float32), "gt": ("S", "H", torch.float32),
        "it_act": ("S", "H", torch.float32), "ft_act": ("S", "H", torch.float32), "st": ("S", "H", torch.float32),
        "ot_act": ("S", "H", torch.float32),
        "attention": ("S", "L", torch.float32), "beta": ("S", 1, torch.float32),
        "context": ("S", "H", torch.float64), "c_hat": ("S", "H", torch.float64),
        "xt": ("T", "2E", torch.float32), "caption_preds": ("T", "V", torch.float64),
        "image_features_before_act": ("L", "H", torch.float32), "average_img_feature": (1, "D", torch.float32),
        "global_img_feature_before_act": (1, "E", torch.float32), "total_static_img_feature": ("L", "H", torch.float32),
    }

    _STATE_SHAPES_GRIDTD = dict(
        [(n, ("S", "H", torch.float64)) for n in ("h1t", "c1t", "g1t", "i1t_act", "f1t_act", "h2t", "c2t", "g2t", "i2t_act",
                                                  "f2t_act", "context", "st", "context_hat", "o1t_act", "o2t_act")] +
        [("attention", ("S", "L", torch.float64)), ("beta", ("S", 1, torch.float64)),
         ("x1t", ("T", "H+2E", torch.float64)), ("x2t", ("T", "2H", torch.float64)),
         ("caption_preds", ("T", "V", torch.float64)),
         ("image_features_before_act", ("L", "H", torch.float32)), ("average_img_feature", (1, "D", torch.float32)),
         ("global_img_feature_before_act", (1, "E", torch.float32)), ("image_features_proj", ("L", "H", torch.float32))])

    def read_state(self, name):
        """Cached decoder array for the images of the last forward: (B, rows, dim)."""
        dims = {"S": self.Tm + 1, "T": self.Tm, "H": self.H, "L": self.L, "V": self.V, "D": self.D, "E": self.E,
                "2E": 2 * self.E, "2H": 2 * self.H, "H+2E": self.H + 2 * self.E, 1: 1}
        r, c, dt = (self._STATE_SHAPES_GRIDTD if self.decoder == "gridtd" else self._STATE_SHAPES)[name]
        out = torch.empty((self.n_images, dims[r], dims[c]), dtype=dt, device=self.device)
        _capi.check(self._lib.lrp_read_state(self._h, name.encode(), C.c_void_p(out.data_ptr()),
                                             out.numel() * out.element_size(), self._stream()))
        return out

    def decoder_explain(self, img_idx, t, variant="sequence", want_attention=True, want_r_words=True):
        n = len(img_idx)
        ii, pi = _i32(img_idx)
        tt, pt = _i32(t)
        R = torch.empty((n, self.L, self.D), dtype=torch.float32, device=self.device)
        att = torch.empty((n, self.L), dtype=torch.float32, device=self.device) if want_attention else None
        rw = torch.empty((n, self.Tm), dtype=torch.float64, device=self.device) if want_r_words else None
        v = _capi.LRP_EXPLAIN_SEQUENCE if variant == "sequence" else _capi.LRP_EXPLAIN_SINGLE_STEP
        _capi.check(self._lib.lrp_decoder_explain(
            self._h, n, pi, pt, v, C.c_void_p(R.data_ptr()),
            C.c_void_p(att.data_ptr()) if att is not None else None,
            C.c_void_p(rw.data_ptr()) if rw is not None else None, self._stream()))
        return R, att, rw

    # ------------------------------------------------------------------ CNN LRP
    def cnn_explain(self, img_idx, R_feat, out=None):
        n = len(img_idx)
        ii, pi = _i32(img_idx)
        R = self._dev(R_feat).reshape(n, self.L, self.D)
        if out is None:
            out = torch.empty((n, self.img_hw[0], self.img_hw[1], 3), dtype=torch.float32, device=self.device)
        _capi.check(self._lib.lrp_cnn_explain(self._h, n, pi, C.c_void_p(R.data_ptr()), C.c_void_p(out.data_ptr()),
                                              self._stream()))
        return out

    # ------------------------------------------------------------------ fine-tune step
    def train_begin(self, lr=2e-4, clipvalue=0.01, beta1=0.9, beta2=0.999, eps=1e-7):
        """Adam(lr, clipvalue) as `ImgCaptioningAdaptiveAttentionLRPInferenceModel.build` compiles it (M:1370);
        returns the layout of the flat parameter / gradient buffers: {name: (offset, size)}."""
        _capi.check(self._lib.lrp_train_begin(self._h, C.c_float(lr), C.c_float(clipvalue), C.c_float(beta1), C.c_float(beta2),
                                              C.c_float(eps)))
        self.train_flat_size = int(self._lib.lrp_train_flat_size(self._h))
        self.train_layout = {}
        for i in range(int(self._lib.lrp_train_num_params(self._h))):
            nm, off, n = C.c_char_p(), C.c_int64(), C.c_int64()
            _capi.check(self._lib.lrp_train_param_info(self._h, i, C.byref(nm), C.byref(off), C.byref(n)))
            self.train_layout[nm.value.decode()] = (int(off.value), int(n.value))
        self.n_images = 0
        return self.train_layout

    def train_set_precision(self, mode):
        """'fp32' (default: fp32-grade gradients) or 'bf16' (BASELINE config 5: the encoder's weight gradients with bf16
        operands, fp32 accumulation and fp32 master weights) — lrp_train_set_precision."""
        m = {"fp32": _capi.LRP_TRAIN_FP32, "bf16": _capi.LRP_TRAIN_BF16}.get(mode)
        if m is None:
            raise ValueError("training precision must be 'fp32' or 'bf16'")
        _capi.check(self._lib.lrp_train_set_precision(self._h, m))
        self.train_precision = mode

    def _train_inputs(self, cap_in, masks, pending=None):
        """Device copies + checks of what the decoder's training forward reads: cap_in (B, T) and the dropout masks.
        pending: what a not yet consumed train_forward converted — the same source OBJECT maps to the same device tensor
        (lrp_train_step insists on the pointers lrp_train_forward read)."""
        masks = masks or {}
        cache = pending or {}
        made = {}
        hit = cache.get("cap_in")
        ci = hit[1] if hit is not None and hit[0] is cap_in else self._dev(cap_in, torch.int32)
        if hit is not None and hit[0] is cap_in:
            ci.record_stream(torch.cuda.current_stream(self.device))
        made["cap_in"] = (cap_in, ci)
        if ci.dim() != 2:
            raise ValueError("cap_in must be (B, T)")
        B, T = ci.shape
        # the kernels index the embedding / the logits with these: validate on the host side of the boundary
        if int(ci.min()) < 0 or int(ci.max()) >= self.V:
            raise ValueError("cap_in holds embedding rows outside [0, V)")
        if B > self.n_images:
            raise RuntimeError("encode_images must run on the batch before the fine-tune step")
        for k in masks:
            if k not in ("image_features", "global", "output", "lstm_in", "lstm_rec", "logits"):
                raise ValueError("unknown dropout mask '%s'" % k)
        self._train_made = made

        def m(key, shape):
            src = masks.get(key)
            if src is None:
                return None
            hit = cache.get(key)
            if hit is not None and hit[0] is src:
                v = hit[1]
                v.record_stream(torch.cuda.current_stream(self.device))      # (made under the early forward's stream)
            else:
                v = self._dev(src)
            made[key] = (src, v)
            if tuple(v.shape) != shape:
                raise ValueError("mask '%s' must be %s" % (key, shape))
            return v
        win = 2 * self.H if self.decoder == "gridtd" else 2 * self.E         # language LSTM input [c_hat | h1] / [emb | glob]
        return ci, (m("image_features", (B, self.L, self.H)), m("global", (B, self.E)), m("output", (B, T, self.H)),
                    m("lstm_in", (T, 4, B, win)), m("lstm_rec", (T, 4, B, self.H)), m("logits", (B, T, self.V)))

    def train_forward(self, cap_in, masks=None):
        """The training-mode decoder forward of `train_step`, ahead of time on the CURRENT stream (it needs neither
        lrp_weight nor the labels): issue it on a side stream under the explanation that produces lrp_weight; the next
        `train_step` with the same (B, T) waits for it and starts at the loss.  Pass the same cap_in / masks to both."""
        ci, (mi, mg, mo, ml, mr, _) = self._train_inputs(cap_in, masks)
        B, T = ci.shape
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        _capi.check(self._lib.lrp_train_forward(self._h, B, T, p(ci), p(mi), p(mg), p(mo), p(ml), p(mr), self._stream()))
        self._train_fwd_keep = self._train_made                             # alive (and reused) until the step has consumed them

    def train_step(self, cap_in, y_idx, lrp_weight, masks=None, grads=None):
        """Gradients of the two-headed loss for the images last encoded.  cap_in / y_idx (B, T) ints (y -1 = no label),
        lrp_weight (B, T, V).  masks: dict with optional 'image_features' (B, L, H), 'global' (B, E), 'output' (B, T, H),
        'lstm_in' (T, 4, B, 2E), 'lstm_rec' (T, 4, B, H) (the LSTM cell's per-gate, per-step dropout).
        Returns (grads flat float32 device tensor, losses (5,) device tensor = total, loss head 1, loss head 2,
        accuracy head 1, accuracy head 2: the list `train_on_batch` returns)."""
        try:
            ci, (mi, mg, mo, ml, mr, mz) = self._train_inputs(cap_in, masks, getattr(self, "_train_fwd_keep", None))
            yi = self._dev(y_idx, torch.int32)
            B, T = ci.shape
            lw = self._dev(lrp_weight).reshape(B, T, self.V)
            if tuple(yi.shape) != (B, T):
                raise ValueError("y_idx must have the shape of cap_in")
            if int(yi.min()) < -1 or int(yi.max()) >= self.V:
                raise ValueError("y_idx holds class indices outside [-1, V)")
            if grads is None:
                grads = torch.empty(self.train_flat_size, dtype=torch.float32, device=self.device)
            losses = torch.empty(5, dtype=torch.float32, device=self.device)
            p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
            _capi.check(self._lib.lrp_train_step(self._h, B, T, p(ci), p(yi), p(lw), p(mi), p(mg), p(mo), p(ml), p(mr), p(mz),
                                                 p(grads), p(losses), self._stream()))
        except Exception:
            # A refused / failed step must not leave the library holding pointers of an early forward whose device tensors
            # are about to lose their keep-alive: torch would hand the same addresses to a retry's fresh masks and the
            # pointer comparison in lrp_train_step would pass for buffers the forward never read.  Drop it first (the
            # stream waits for the early forward), THEN release the tensors.
            self._lib.lrp_train_drop_forward(self._h, self._stream())
            self._train_fwd_keep = None
            raise
        self._train_fwd_keep = None                                         # consumed: the step's work is stream-ordered behind it
        return grads, losses

    def train_apply(self, grads):
        """Adam update of the master weights from the (all-reduced) flat gradient; cached images are dropped."""
        _capi.check(self._lib.lrp_train_apply(self._h, C.c_void_p(grads.data_ptr()), self._stream()))
        self.n_images = 0
        self.captions = None

    def train_weights_device(self):
        """Master weights as {name: flat float32 device view} of one fresh copy of the flat buffer."""
        flat = torch.empty(self.train_flat_size, dtype=torch.float32, device=self.device)
        _capi.check(self._lib.lrp_train_get_master(self._h, C.c_void_p(flat.data_ptr()), self._stream()))
        return {k: flat[o:o + n] for k, (o, n) in self.train_layout.items()}

    def train_weights(self):
        """Master weights as {name: float32 ndarray} (checkpointing / tests)."""
        return {k: v.cpu().numpy().copy() for k, v in self.train_weights_device().items()}

    # ------------------------------------------------------------------ caption generation
    def gen_begin(self, n_rows):
        """Start an incremental decode over the first n_rows cached feature slots (one hypothesis per row)."""
        _capi.check(self._lib.lrp_decoder_gen_begin(self._h, int(n_rows), self._stream()))
        self._gen_rows = int(n_rows)

    def gen_step(self, step, parent=None, word=None):
        """One decoder step for every row; row r continues row parent[r] with tokenizer id word[r] appended
        (step 0: SOS).  Returns the (n_rows, V) float64 logits of position `step` on the device."""
        n = self._gen_rows
        out = torch.empty((n, self.V), dtype=torch.float64, device=self.device)
        if step > 0:
            pp, ppi = _i32(parent)
            ww, wwi = _i32(word)
        else:
            ppi = wwi = None
        _capi.check(self._lib.lrp_decoder_gen_step(self._h, n, ppi, wwi, int(step), C.c_void_p(out.data_ptr()), self._stream()))
        return out

    def log_softmax_topk(self, logits, k):
        """`_log_softmax` + `np.argpartition(preds, -k)[:, -k:]` (explainers.py:45-48, :76-78) on the device for the
        (rows, V) float64 logits of `gen_step`: (ids (rows, k) int32 model columns, logp (rows, k) float64), both by
        descending probability.  Only these k pairs per row have to cross PCIe."""
        return log_softmax_topk(logits, k)

    # ------------------------------------------------------------------ gradient baselines
    WALKS = {"lrp": 0, "gradient": 1, "input_x_gradient": 2, "guided_backprop": 3}

    def decoder_gradient(self, img_idx, t, want_r_words=True):
        """_lstm_decoder_backward (explainers.py:780-832 / :1452-1532) for n (image, t) units: (n, L, D) float32
        and the per-word sums r_words (n, Tm) float64 (columns >= t are zero)."""
        n = len(img_idx)
        ii, pi = _i32(img_idx)
        tt, pt = _i32(t)
        d = torch.empty((n, self.L, self.D), dtype=torch.float32, device=self.device)
        rw = torch.zeros((n, self.Tm), dtype=torch.float64, device=self.device) if want_r_words else None
        _capi.check(self._lib.lrp_decoder_gradient(self._h, n, pi, pt, C.c_void_p(d.data_ptr()),
                                                   C.c_void_p(rw.data_ptr()) if rw is not None else None, self._stream()))
        return d, rw

    def cnn_walk(self, img_idx, head, walk="gradient", out=None):
        """Gradient / InputTimesGradient / GuidedBackprop `.analyze([X, head])` (gradient_based.py:101-265) on the
        cached forward of `lrp_encode_images`; walk='lrp' is cnn_explain."""
        n = len(img_idx)
        ii, pi = _i32(img_idx)
        Hd = self._dev(head).reshape(n, self.L, self.D)
        if out is None:
            out = torch.empty((n, self.img_hw[0], self.img_hw[1], 3), dtype=torch.float32, device=self.device)
        _capi.check(self._lib.lrp_cnn_walk(self._h, n, pi, C.c_void_p(Hd.data_ptr()), C.c_void_p(out.data_ptr()),
                                           self.WALKS[walk], self._stream()))
        return out

    def explain_tokens(self, img_idx, t, variant="sequence", out=None, want_R_feat=False, want_attention=False,
                       want_r_words=False):
        """Fused decoder-LRP -> CNN-LRP for n (image, t) pairs: (n,H,W,3) heat-map relevances."""
        n = len(img_idx)
        ii, pi = _i32(img_idx)
        tt, pt = _i32(t)
        if out is None:
            out = torch.empty((n, self.img_hw[0], self.img_hw[1], 3), dtype=torch.float32, device=self.device)
        R = torch.empty((n, self.L, self.D), dtype=torch.float32, device=self.device) if want_R_feat else None
        att = torch.empty((n, self.L), dtype=torch.float32, device=self.device) if want_attention else None
        rw = torch.empty((n, self.Tm), dtype=torch.float64, device=self.device) if want_r_words else None
        v = _capi.LRP_EXPLAIN_SEQUENCE if variant == "sequence" else _capi.LRP_EXPLAIN_SINGLE_STEP
        p = lambda x: C.c_void_p(x.data_ptr()) if x is not None else None
        _capi.check(self._lib.lrp_explain_tokens(self._h, n, pi, pt, v, p(out), p(R), p(att), p(rw), self._stream()))
        return out, R, att, rw

    def set_precision(self, mode):
        """'bf16x3' (default: split-bf16 walk, hi*hi' + hi*lo' + lo*hi' in every layer, 16 mantissa bits on both operands —
        parity independent of the weight statistics), 'fp32' (exact fp32 MFMA, the reference's arithmetic), 'f16x2'
        (opt-in fast mode: fp16-pair relevance, ONE fp16 per weight below the top block — two MFMAs per product; its error
        depends on how concentrated the weights are, see include/lrp_hip.h) or 'bf16x3_fast' (split forward activations
        too).  A mode change drops the encode caches: call encode_images again."""
        m = {"fp32": _capi.LRP_PREC_FP32, "bf16x3": _capi.LRP_PREC_BF16X3, "bf16x3_fast": _capi.LRP_PREC_BF16X3_FAST,
             "f16x2": _capi.LRP_PREC_F16X2}.get(mode)
        if m is None:
            raise ValueError("precision must be 'fp32', 'bf16x3', 'bf16x3_fast' or 'f16x2'")
        _capi.check(self._lib.lrp_set_precision(self._h, m))
        if mode != self.precision:
            self.n_images = 0                            # the library dropped the caches of the other arithmetic
            self.captions = None
        self.precision = mode

    def set_fast_layers(self, mask):
        """Which conv layers take the two-MFMA form in 'f16x2' mode (lrp_set_fast_layers): bit li of `mask`, an iterable of
        layer indices, or None / -1 for the built-in rule.  0 = fp16 pairs with three MFMAs in every layer.  A change drops
        the encode caches while the engine is in 'f16x2' mode.  See calibration.calibrate_fast_mode for a measured choice."""
        if mask is None:
            mask = -1
        elif not isinstance(mask, int):
            mask = sum(1 << int(li) for li in set(mask))
        _capi.check(self._lib.lrp_set_fast_layers(self._h, int(mask)))
        if self.precision == "f16x2" and mask != getattr(self, "fast_layers", -1):
            self.n_images = 0
            self.captions = None
        self.fast_layers = mask

    # ------------------------------------------------------------------ profiling hooks (bench.py)
    def profile_enable(self, on=True):
        _capi.check(self._lib.lrp_profile_enable(self._h, int(bool(on))))

    def profile_query(self):
        n, ms, fl = C.c_int64(), C.c_double(), C.c_double()
        _capi.check(self._lib.lrp_profile_query(self._h, C.byref(n), C.byref(ms), C.byref(fl)))
        return n.value, ms.value, fl.value


def _profile_records(self, cap=256):
    ms = (C.c_double * cap)()
    fl = (C.c_double * cap)()
    n = C.c_int32()
    _capi.check(self._lib.lrp_profile_records(self._h, cap, ms, fl, C.byref(n)))
    return [(ms[i], fl[i]) for i in range(n.value)]


LRPEngine.profile_records = _profile_records


CONV_SPLIT_BF16 = 0x100


def op_conv(x, w_hwio, bias, aux, mode, taps=9, split_bf16=False):
    """Operator-level entry for unit tests of the MFMA conv kernel (see lrp_op_conv); split_bf16 runs the
    same operator on the split-bf16 path (LRP_CONV_SPLIT_BF16)."""
    lib = _capi.load()
    dev = x.device
    w = np.ascontiguousarray(w_hwio, dtype=np.float32)
    Cin, Cout = w.shape[2], w.shape[3]
    NB, H, W, _ = x.shape
    bwd = mode >= 2
    if mode == 3:
        out = torch.empty((NB, 2 * H, 2 * W, Cin), dtype=torch.float32, device=dev)
    elif bwd:
        out = torch.empty((NB, H, W, Cin), dtype=torch.float32, device=dev)
    else:
        out = torch.empty((NB, H, W, Cout), dtype=torch.float32, device=dev)
    b = np.ascontiguousarray(bias, dtype=np.float32) if bias is not None else None
    _capi.check(lib.lrp_op_conv(C.c_void_p(x.data_ptr()), w.ctypes.data_as(C.c_void_p),
                                b.ctypes.data_as(C.c_void_p) if b is not None else None,
                                C.c_void_p(aux.data_ptr()) if aux is not None else None,
                                C.c_void_p(out.data_ptr()), NB, H, W, Cin, Cout, taps,
                                mode | (CONV_SPLIT_BF16 if split_bf16 else 0),
                                C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    return out


def op_conv_pool_sparse(sc, pos, w_hwio, gate, reps=1):
    """The conv-LRP launch behind a 2x2 max-pool on the 2:4-sparse matrix cores (lrp_op_conv_pool_sparse): sc (NB, Hp, Wp, Cout)
    float32 = each window's one non-zero, pos (same shape) uint8 = its position 2 dy + dx, w_hwio (3, 3, Cin, Cout) numpy,
    gate (NB, 2 Hp, 2 Wp, Cin) -> (NB, 2 Hp, 2 Wp, Cin) float32."""
    lib = _capi.load()
    w = np.ascontiguousarray(w_hwio, dtype=np.float32)
    Cin, Cout = w.shape[2], w.shape[3]
    NB, Hp, Wp, _ = sc.shape
    assert pos.dtype == torch.uint8 and tuple(pos.shape) == tuple(sc.shape) and tuple(gate.shape) == (NB, 2 * Hp, 2 * Wp, Cin)
    sc, pos, gate = sc.contiguous(), pos.contiguous(), gate.contiguous()
    out = torch.empty((NB, 2 * Hp, 2 * Wp, Cin), dtype=torch.float32, device=sc.device)
    _capi.check(lib.lrp_op_conv_pool_sparse(C.c_void_p(sc.data_ptr()), C.c_void_p(pos.data_ptr()), w.ctypes.data_as(C.c_void_p),
                                            C.c_void_p(gate.data_ptr()), C.c_void_p(out.data_ptr()), NB, Hp, Wp, Cin, Cout, int(reps),
                                            _cur_stream(sc.device)))
    return out


def op_sgemm(A, B, transA=False, transB=False, C_init=None, split=True):
    """C (+)= op(A) op(B) through lrp_op_sgemm (fp32 MFMA); A, B 2-D device tensors (views with a row stride allowed)."""
    lib = _capi.load()
    M, K = (A.shape[1], A.shape[0]) if transA else A.shape
    N = B.shape[0] if transB else B.shape[1]
    out = torch.empty((M, N), dtype=torch.float32, device=A.device) if C_init is None else C_init
    ws = torch.empty(8 << 20, dtype=torch.float32, device=A.device) if split else None
    _capi.check(lib.lrp_op_sgemm(C.c_void_p(A.data_ptr()), C.c_void_p(B.data_ptr()), C.c_void_p(out.data_ptr()), M, N, K,
                                 A.stride(0), B.stride(0), out.stride(0), int(transA), int(transB), int(C_init is not None),
                                 C.c_void_p(ws.data_ptr()) if ws is not None else None, ws.numel() if ws is not None else 0,
                                 _cur_stream(A.device)))
    return out


def op_conv_wgrad(x, dz, bf16=False):
    """Weight / bias gradient of a 3x3 'same' conv (lrp_op_conv_wgrad[_bf16]): x (NB,H,W,Cin), dz (NB,H,W,Cout) -> (dw HWIO, db)."""
    lib = _capi.load()
    NB, H, W, Cin = x.shape
    Cout = dz.shape[3]
    dw = torch.empty((3, 3, Cin, Cout), dtype=torch.float32, device=x.device)
    db = torch.empty((Cout,), dtype=torch.float32, device=x.device)
    ws = torch.empty(16 << 20, dtype=torch.float32, device=x.device)
    fn = lib.lrp_op_conv_wgrad_bf16 if bf16 else lib.lrp_op_conv_wgrad
    _capi.check(fn(C.c_void_p(x.data_ptr()), C.c_void_p(dz.data_ptr()), C.c_void_p(dw.data_ptr()),
                                      C.c_void_p(db.data_ptr()), NB, H, W, Cin, Cout, C.c_void_p(ws.data_ptr()), ws.numel(),
                                      _cur_stream(x.device)))
    return dw, db


def _cur_stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def op_epsilon_dense(x, W, R, epsilon):
    """EpsilonRule(bias=False) for a Dense layer (RR:113-144): x (N,Din), W (Din,Dout) numpy, R (N,Dout)."""
    lib = _capi.load()
    W = np.ascontiguousarray(W, dtype=np.float32)
    out = torch.empty_like(x)
    _capi.check(lib.lrp_op_epsilon_dense(C.c_void_p(x.data_ptr()), W.ctypes.data_as(C.c_void_p), C.c_void_p(R.data_ptr()),
                                         C.c_void_p(out.data_ptr()), x.shape[0], W.shape[0], W.shape[1], float(epsilon),
                                         _cur_stream(x.device)))
    return out


def op_batchnorm_lrp(x, gamma, beta, mean, var, bn_eps, R):
    """BatchNormalizationReverseLayer (RA:197-257), channels-last tensors on the GPU."""
    lib = _capi.load()
    out = torch.empty_like(x)
    p = lambda t: C.c_void_p(t.data_ptr())
    _capi.check(lib.lrp_op_batchnorm_lrp(p(x), p(gamma), p(beta), p(mean), p(var), float(bn_eps), p(R), p(out), x.numel(),
                                         x.shape[-1], _cur_stream(x.device)))
    return out


def op_add_lrp(a, b, R):
    """AddReverseLayer (RA:260-286)."""
    lib = _capi.load()
    Ra, Rb = torch.empty_like(a), torch.empty_like(b)
    p = lambda t: C.c_void_p(t.data_ptr())
    _capi.check(lib.lrp_op_add_lrp(p(a), p(b), p(R), p(Ra), p(Rb), a.numel(), _cur_stream(a.device)))
    return Ra, Rb


def op_avgpool_lrp(x, R, k):
    """AveragePoolingReverseLayer (RA:289-316), k x k / stride k average pooling, channels-last."""
    lib = _capi.load()
    x, R = x.contiguous(), R.contiguous()
    NB, H, W, Cc = x.shape
    out = torch.empty_like(x)
    p = lambda t: C.c_void_p(t.data_ptr())
    _capi.check(lib.lrp_op_avgpool_lrp(p(x), p(R), p(out), NB, H, W, Cc, int(k), _cur_stream(x.device)))
    return out


def log_softmax_topk(logits, k):
    """lrp_op_log_softmax_topk: logits (rows, V) float64 device tensor -> (ids int32 (rows, k), logp float64 (rows, k))."""
    lib = _capi.load()
    x = logits.contiguous()
    if x.dtype != torch.float64 or x.dim() != 2:
        raise ValueError("expected a (rows, V) float64 tensor")
    rows, V = x.shape
    ids = torch.empty((rows, k), dtype=torch.int32, device=x.device)
    lp = torch.empty((rows, k), dtype=torch.float64, device=x.device)
    _capi.check(lib.lrp_op_log_softmax_topk(C.c_void_p(x.data_ptr()), rows, V, int(k), C.c_void_p(ids.data_ptr()),
                                            C.c_void_p(lp.data_ptr()), _cur_stream(x.device)))
    return ids, lp


_LUT_CACHE = {}


def heatmap_render(R_img, gamma=0.95, color_conversion=None):
    """`heatmap(postprocess(relevance, color_conversion))` of the harness (explain_image.py:55-60) on the device:
    R_img (n, H, W, 3) relevance tensor -> (n, H, W, 3) float32 RGB in [0, 1] (seismic colormap)."""
    from .postprocess import _seismic
    lib = _capi.load()
    R = R_img.contiguous()
    if color_conversion in ("RGBtoBGR", "BGRtoRGB"):
        R = R.flip(-1).contiguous()              # (the channel sum does not care, the gamma maximum neither)
    n, Hh, Ww, Cc = R.shape
    key = str(R.device)
    if key not in _LUT_CACHE:
        _LUT_CACHE[key] = torch.as_tensor(np.ascontiguousarray(_seismic(np.arange(256)), dtype=np.float32)).to(R.device)
    out = torch.empty((n, Hh, Ww, 3), dtype=torch.float32, device=R.device)
    _capi.check(lib.lrp_heatmap_render(C.c_void_p(R.data_ptr()), C.c_void_p(_LUT_CACHE[key].data_ptr()),
                                       C.c_void_p(out.data_ptr()), n, Hh * Ww, Cc, C.c_float(gamma), _cur_stream(R.device)))
    return out


class switches(object):
    """`with switches(LRP_UP2_PW=0, ...):` — set the library's A/B switches for the body and restore the
    environment afterwards (lrp_reload_switches re-reads them: they are not looked up on the launch path).  For tests and
    measurement scripts; a switch that changes the encode caches takes effect at the next encode_images."""

    def __init__(self, **env):
        self.env = {k: str(v) for k, v in env.items()}
        self.old = {}

    def __enter__(self):
        import os
        for k, v in self.env.items():
            if not k.startswith("LRP_"):
                raise ValueError("not a library switch: %s" % k)
            self.old[k] = os.environ.get(k)
            os.environ[k] = v
        _capi.check(_capi.load().lrp_reload_switches())
        return self

    def __exit__(self, *exc):
        import os
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        _capi.check(_capi.load().lrp_reload_switches())
        return False


def preprocess_images(rgb_u8, size=(224, 224)):
    """models/preprocessors.py:38-53 on the device: (NB, H0, W0, 3) uint8 RGB tensor -> (NB, H, W, 3) float32 BGR,
    mean-subtracted, nearest-neighbour resized like keras `load_img(target_size=size)`."""
    lib = _capi.load()
    x = rgb_u8.contiguous()
    if x.dtype != torch.uint8 or x.dim() != 4 or x.shape[-1] != 3:
        raise ValueError("expected a (NB, H0, W0, 3) uint8 tensor")
    NB, H0, W0, _ = x.shape
    out = torch.empty((NB, size[0], size[1], 3), dtype=torch.float32, device=x.device)
    _capi.check(lib.lrp_preprocess_images(C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), NB, H0, W0, size[0], size[1],
                                          _cur_stream(x.device)))
    return out


def heatmap_scores(R_img, mode):
    """LRP-inference score per heat-map (model.py:1675-1686) on the device: R_img (n,H,W,C) -> (n,) float64."""
    lib = _capi.load()
    m = {"mean": 0, "pos_mean": 1, "quantile": 2}.get(mode)
    if m is None:
        raise NotImplementedError("the lrp inference mode is not available")
    R = R_img.contiguous()
    n, C_ = R.shape[0], R.shape[-1]
    out = torch.empty((n,), dtype=torch.float64, device=R.device)
    _capi.check(lib.lrp_heatmap_scores(C.c_void_p(R.data_ptr()), C.c_void_p(out.data_ptr()), n, R[0].numel() // C_, C_, m,
                                       _cur_stream(R.device)))
    return out
