"""Host side of the LRP-inference fine-tune loop (train.py:519-593, `TrainingLRPInferenceAdaptive.run`): per batch

    y_pred     = keras_model.predict_on_batch(X + [zeros])[0]          train.py:573
    lrp_weight = LRPInferenceLayerAdaptive(...).call(X + [y_pred])     train.py:574-576
    losses     = keras_model.train_on_batch(X + [lrp_weight], [y, y])  train.py:579

with the three calls mapped onto one `lrp_handle`: the encoder runs once per batch (the reference runs it for the
predict, once per explained word, and again for the training forward), the explanation and the gradient step share
its caches, and the update is liblrp_hip.so's Adam (lrp_train_apply).  Data parallel: one process per GPU, each on
its shard of the batch; the flat gradient is averaged with ONE all-reduce (RCCL over xGMI with backend "nccl").
"""
import numpy as np
import torch

from .lrp_inference import LRPInferenceLayerAdaptive, LRPInferenceLayergridTD
from .parallel import average_gradients


class TrainingLRPInferenceAdaptive(object):
    _LAYER = LRPInferenceLayerAdaptive
    _CLIPVALUE = 0.01                                  # models/model.py:1370

    def __init__(self, explainer, learning_rate=2e-4, clipvalue=None, drop_rate=0.5, lrp_inference_mode="mean", stop_words=(),
                 seed=0, process_group=None):
        """explainer: an `ExplainImgCaptioningAdaptiveAttention` (its engine holds the weights).  Optimiser as compiled at
        models/model.py:1370 (`Adam(lr, clipvalue=0.01)`), dropout rate `config.drop_rate` (config.py:16) on the
        image_features / global_img_feature / decoder output Dropout layers (M:1348, :1352, :1363) and inside the LSTM
        cell (`dropout` / `recurrent_dropout`, M:1356-1358)."""
        self._explainer = explainer
        self._engine = explainer._engine
        self._lrp_layer = self._LAYER(explainer, lrp_inference_mode, stop_words)
        clipvalue = self._CLIPVALUE if clipvalue is None else clipvalue
        self._drop_rate = float(drop_rate)
        self._gen = torch.Generator(device=self._engine.device)
        self._gen.manual_seed(int(seed))
        self._pg = process_group
        self.layout = self._engine.train_begin(lr=learning_rate, clipvalue=clipvalue)
        self._grads = torch.zeros(self._engine.train_flat_size, dtype=torch.float32, device=self._engine.device)
        self._side = torch.cuda.Stream(device=self._engine.device)

    # -- keras_model.predict_on_batch(X + [zeros])[0]: teacher-forced logits (B, T, V), inference mode
    def predict_on_batch(self, X):
        captions_input, imgs = X[0], X[1]
        eng = self._engine
        cap_in = np.asarray(captions_input, dtype=np.int64)
        B, T = cap_in.shape
        eng.encode_images(imgs)
        # the replay feeds SOS, then caption[i-1]: hand it the input row shifted by one (ids = embedding row + 1)
        caps = [[int(c) + 1 for c in cap_in[b, 1:]] + [int(eng.cfg.eos_id)] for b in range(B)]
        if not np.all(cap_in[:, 0] + 1 == int(eng.cfg.sos_id)):
            raise ValueError("captions_input must start with the start-of-sentence token")
        eng.decoder_forward(caps)
        return self._logits(B, T)

    def _logits(self, B, T):
        return self._engine.read_state("caption_preds")[:B, :T].to(torch.float32)

    def _masks(self, B, T):
        p = self._drop_rate
        if not 0.0 < p < 1.0:
            return None
        eng = self._engine
        mk = lambda *s: (torch.rand(*s, device=eng.device, generator=self._gen) >= p).to(torch.float32) / (1.0 - p)
        masks = {"image_features": mk(B, eng.L, eng.H), "global": mk(B, eng.E), "output": mk(B, T, eng.H),
                 "lstm_in": mk(T, 4, B, 2 * eng.E), "lstm_rec": mk(T, 4, B, eng.H)}
        if eng.decoder == "gridtd":
            masks["logits"] = mk(B, T, eng.V)          # Dropout on the logits (M:1303-1304)
        return masks

    def train_on_batch(self, X, y, lrp_weight=None):
        """One iteration of the `while True` body (train.py:571-580).  X = [captions_input (B, T), images (B, H, W, 3)],
        y (B, T, V) one-hot (all-zero rows = padding) or (B, T) class indices with -1 for padding.
        Returns [loss, loss_head1, loss_head2, acc_head1, acc_head2] like `train_on_batch`."""
        eng = self._engine
        cap_in = np.asarray(X[0], dtype=np.int32)
        B, T = cap_in.shape
        y = np.asarray(y)
        if y.ndim == 3:
            y_idx = np.where(y.sum(-1) > 0, y.argmax(-1), -1).astype(np.int32)
        else:
            y_idx = y.astype(np.int32)
        masks = self._masks(B, T)
        cap_dev = torch.as_tensor(cap_in, dtype=torch.int32).to(eng.device)
        if lrp_weight is None:
            y_pred = self.predict_on_batch(X)
            # the training-mode decoder forward needs only the features: run it on a side stream under the explanation
            cur = torch.cuda.current_stream(eng.device)
            self._side.wait_stream(cur)
            with torch.cuda.stream(self._side):
                eng.train_forward(cap_dev, masks)
            lrp_weight = self._lrp_layer.call_device(X[1], y_pred, images_encoded=True)
            if eng.n_images < B:                       # (cannot happen after predict_on_batch; keeps the caches' owner explicit)
                eng.encode_images(X[1])
        else:
            eng.encode_images(X[1])
        grads, losses = eng.train_step(cap_dev, y_idx, lrp_weight, masks, grads=self._grads)
        grads, losses = average_gradients(grads, losses, self._pg)    # one bucket: the whole flat gradient
        eng.train_apply(grads)
        return [float(v) for v in losses.cpu().numpy()]

    def get_weights(self):
        """Master weights, shaped like the arrays the model was built from (conv HWIO, dense (in, out))."""
        shapes = {k: np.shape(v) for k, v in self._explainer._model.weights.items()}
        return {k: v.reshape(shapes.get(k, v.shape)) for k, v in self._engine.train_weights().items()}

    def save_weights(self, path):
        """`keras_model.save_weights(...)` of the loop (train.py:585-587): an `.npz` bundle of the current master
        weights that the explainer classes take back as `weight_path` (hdf5 is not available here)."""
        np.savez(path, **self.get_weights())
        return path


class TrainingLRPInferenceGridTD(TrainingLRPInferenceAdaptive):
    """train.py:596-669 on ImgCaptioningGridTDLRPInferenceModel (models/model.py:1254-1311): the same loop body over the
    grid-TD captioner — `Adam(lr, clipvalue=0.1)` (M:1307), a Dropout on the logits as well (M:1303-1304), dropout
    inside the language LSTM.  `predict_on_batch` returns the Keras model's logits, (h2 + c_hat) W + b (M:816), not the
    explainer's h2-only replay (models/explainers.py:1154)."""
    _LAYER = LRPInferenceLayergridTD
    _CLIPVALUE = 0.1

    def _logits(self, B, T):
        from .engine import op_sgemm
        eng = self._engine
        w = eng.train_weights_device()
        Wout = w["output_W"].view(eng.H, eng.V)
        h2 = eng.read_state("h2t")[:B, 1:T + 1].to(torch.float32).reshape(B * T, eng.H).contiguous()
        ch = eng.read_state("context_hat")[:B, 1:T + 1].to(torch.float32).reshape(B * T, eng.H).contiguous()
        out = op_sgemm(h2, Wout, split=False)
        out = op_sgemm(ch, Wout, C_init=out, split=False)
        return (out + w["output_b"]).view(B, T, eng.V)
