"""Batched driver of the hot path used during LRP-inference fine-tuning
(models/model.py:1379-1691 LRPInferenceLayerAdaptive, call() at :1641-1691):
for every image and every non-stop-word position of its predicted caption one
heat-map -> one scalar score -> lrp_weight[b, i, word] = 1 + score.

The reference loops image by image, word by word (B x <=20 sequential LRP passes);
here all (image, position) units of the batch go through ONE explain call."""
import numpy as np


class LRPInferenceLayerAdaptive(object):
    def __init__(self, explainer, lrp_inference_mode="mean", stop_words=(), color_conversion="BGRtoRGB"):
        if lrp_inference_mode not in ("mean", "pos_mean", "quantile"):
            raise NotImplementedError("the lrp inference mode is not available")      # model.py:1685-1686
        self._explainer = explainer
        self._engine = explainer._engine
        self._preprocessor = explainer._preprocessor
        self._EOS_ENCODED = self._preprocessor.EOS_TOKEN_LABEL_ENCODED
        self._lrp_inference_mode = lrp_inference_mode
        self._stop_words = set(stop_words)
        self._color_conversion = color_conversion

    def _positions(self, caption_encoded):
        """model.py:1664-1671: skip stop words, stop at EOS."""
        out = []
        for i, w in enumerate(caption_encoded):
            word = self._preprocessor._word_of.get(int(w)) if getattr(self._preprocessor, "_word_of", None) else None
            if word is not None and word in self._stop_words:
                continue
            if int(w) == self._EOS_ENCODED:
                break
            out.append(i)
        return out

    def _plan(self, words, T):
        """words (B, T) predicted tokenizer ids -> (captions to replay, [(b, position, word)] to explain)."""
        eng = self._engine
        caps, pairs = [], []
        for b in range(len(words)):
            cap = np.asarray(words[b]).astype(np.int64)
            # the replay needs a caption that ends in EOS: cut at the first EOS (positions after it are never explained)
            eos = np.where(cap == self._EOS_ENCODED)[0]
            n = int(eos[0]) + 1 if len(eos) else T
            c = [int(x) for x in cap[:n]]
            if c[-1] != self._EOS_ENCODED:
                c = c[:eng.Tm - 1] + [self._EOS_ENCODED]
            caps.append(c)
            for i in self._positions(cap):
                if i < len(c):
                    pairs.append((b, i, int(cap[i])))
        return caps, pairs

    def call(self, inputs, images_encoded=False):
        """images_encoded: the engine already holds these images (the fine-tune loop encodes once per batch)."""
        assert len(inputs) == 3
        _, img_inputs, y_preds = inputs
        y_preds = np.asarray(y_preds)
        B, T, V = y_preds.shape
        eng = self._engine
        if B > eng.max_images:
            raise ValueError("batch larger than the engine's max_images")
        caps, pairs = self._plan(np.argmax(y_preds, axis=-1) + 1, T)                # model.py:1661-1662
        out = np.zeros(y_preds.shape, dtype=np.float64)
        if pairs:
            if not (images_encoded and eng.n_images == B):
                eng.encode_images(img_inputs)
            eng.decoder_forward(caps)
            for lo in range(0, len(pairs), eng.max_tokens):
                chunk = pairs[lo:lo + eng.max_tokens]
                R, _, _, _ = eng.explain_tokens([p[0] for p in chunk], [p[1] + 1 for p in chunk])
                scores = self._scores(R)
                for (b, i, w), s in zip(chunk, scores):
                    if w < V:
                        out[b, i, w] = s                                            # model.py:1687 (index = tokenizer id)
        return 1 + out                                                              # model.py:1690

    def call_device(self, img_inputs, y_preds_dev, images_encoded=False):
        """`call` with the (B, T, V) logits and the result on the device (float32 tensor): the fine-tune loop never
        moves a vocabulary-sized array over PCIe — only the (B, T) arg-max words come to the host."""
        import torch
        from .engine import heatmap_scores
        eng = self._engine
        B, T, V = y_preds_dev.shape
        if B > eng.max_images:
            raise ValueError("batch larger than the engine's max_images")
        caps, pairs = self._plan((y_preds_dev.argmax(dim=-1) + 1).cpu().numpy(), T)
        lw = torch.ones((B, T, V), dtype=torch.float32, device=y_preds_dev.device)
        if pairs:
            if not (images_encoded and eng.n_images == B):
                eng.encode_images(img_inputs)
            eng.decoder_forward(caps)
            for lo in range(0, len(pairs), eng.max_tokens):
                chunk = [p for p in pairs[lo:lo + eng.max_tokens]]
                R, _, _, _ = eng.explain_tokens([p[0] for p in chunk], [p[1] + 1 for p in chunk])
                s = heatmap_scores(R, self._lrp_inference_mode)
                keep = [k for k, p in enumerate(chunk) if p[2] < V]
                if keep:
                    idx = torch.as_tensor([[chunk[k][0], chunk[k][1], chunk[k][2]] for k in keep], device=lw.device)
                    lw[idx[:, 0], idx[:, 1], idx[:, 2]] = 1 + s[torch.as_tensor(keep, device=lw.device)].to(torch.float32)
        return lw

    def _scores(self, R):
        """model.py:1675-1686 for the device tensor R (n,H,W,3), in liblrp_hip.so (lrp_heatmap_scores)."""
        from .engine import heatmap_scores
        return heatmap_scores(R, self._lrp_inference_mode).cpu().numpy()


class LRPInferenceLayergridTD(LRPInferenceLayerAdaptive):
    """models/model.py:1693-2062 (call() at :2013-2062): the same batch driver over the grid-TD engine — the
    reference duplicates the whole class; here only the explainer handed in differs
    (an `ExplainImgCaptioningGridTDModel`)."""

    def __init__(self, explainer, lrp_inference_mode="mean", stop_words=(), color_conversion="BGRtoRGB"):
        if getattr(explainer, "_decoder_kind", None) != "gridtd":
            raise ValueError("LRPInferenceLayergridTD needs a grid-TD explainer")
        super(LRPInferenceLayergridTD, self).__init__(explainer, lrp_inference_mode, stop_words, color_conversion)
