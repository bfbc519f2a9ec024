"""`LRPSequentialPresetA` with the iNNvestigate call surface the captioning code uses
(innvestigate/analyzer/relevance_based/relevance_analyzer.py:695-721, base.py:328-347, :478-520):

    analyzer = LRPSequentialPresetA(image_model, epsilon=0.01, neuron_selection_mode='replace')
    relevance = analyzer.analyze([X, R])          # X (N,224,224,3), R (N,14,14,512) -> (N,224,224,3)

`image_model` is an `ImageModelSpec` (layer list + weights) instead of a Keras model: the
symbolic graph reversal of the reference (utils/keras/graph.py:704-942) is not reproduced,
only its semantics — Conv -> Alpha1Beta0Rule, MaxPooling -> gradient routing, fused ReLU ->
pass-through, head relevance := second input — executed by the HIP encoder path."""
import numpy as np

from .engine import LRPEngine
from .explainers import _EngineAnalyzer
from .synthetic import VGG16_CFG


class NotAnalyzeableModelException(Exception):
    """innvestigate/analyzer/base.py:37-39."""


class ImageModelSpec(object):
    """The truncated encoder `Model(input_1 -> block5_conv3)` (explainers.py:29-30) as data: a VGG-style conv list,
    or (resnet=dict(stem, stacks)) the ResNet-v1 bottleneck stack cut at conv5_block3_out.  The analyzer's head
    shape follows from it — (14,14,512) / (7,7,2048) in the reference's hard-coded table (base.py:370-373)."""

    def __init__(self, weights, cnn_cfg=VGG16_CFG, img_hw=(224, 224), resnet=None):
        self.img_hw = tuple(img_hw)
        self.resnet = resnet
        if resnet is None:
            self.cnn_cfg = list(cnn_cfg)
            self.weights = {k: v for k, v in weights.items() if k.endswith(("_W", "_b")) and k[:-2] in {c[0] for c in cnn_cfg}}
            missing = [c[0] for c in cnn_cfg if c[0] + "_W" not in self.weights or c[0] + "_b" not in self.weights]
        else:
            from .synthetic import resnet_conv_list
            self.cnn_cfg = VGG16_CFG
            names = [c[0] for c in resnet_conv_list(resnet["stacks"], resnet.get("stem", 64))]
            sufs = ("_conv_W", "_conv_b", "_bn_gamma", "_bn_beta", "_bn_mean", "_bn_var")
            self.weights = {n + s: weights[n + s] for n in names for s in sufs if n + s in weights}
            missing = [n + s for n in names for s in sufs if n + s not in weights]
        if missing:
            raise NotAnalyzeableModelException("weights missing for layers %s" % missing[:8])

    def output_shape(self):
        h, w = self.img_hw
        if self.resnet is not None:
            stacks = self.resnet["stacks"]
            d = 4 * 2 ** (len(stacks) - 1)
            return h // d, w // d, 4 * stacks[-1][0]
        for _, _, _, pool in self.cnn_cfg:
            if pool:
                h, w = h // 2, w // 2
        return h, w, self.cnn_cfg[-1][2]


class LRPSequentialPresetA(object):
    def __init__(self, model, epsilon=0.1, neuron_selection_mode="replace", max_batch=8, device=None, **kwargs):
        if neuron_selection_mode not in ["max_activation", "index", "all", "replace"]:
            raise ValueError("neuron_selection parameter is not valid.")             # base.py:332-333
        if neuron_selection_mode != "replace":
            raise NotImplementedError("only neuron_selection_mode='replace' is on the captioning hot path")
        if not (epsilon > 0):
            raise ValueError("epsilon must be > 0")                                   # relevance_based/utils.py:52-60
        self._epsilon = epsilon          # used by Dense layers only (PresetA); the truncated encoders have none
        self._neuron_selection_mode = neuron_selection_mode
        self._model = model
        h, w, c = model.output_shape()
        # the decoder half of the handle is idle here: minimal dims
        self._engine = LRPEngine(decoder="adaptive", cnn_cfg=model.cnn_cfg, img_hw=model.img_hw, L=h * w, D=c, H=4, E=4,
                                 V=4, max_images=max_batch, max_tokens=max_batch, max_caption_len=2, device=device,
                                 resnet=model.resnet)
        self._engine.set_weights(model.weights)
        self._impl = _EngineAnalyzer(self._engine)

    def analyze(self, X, neuron_selection=None):
        if neuron_selection is not None:
            raise ValueError("Only neuron_selection_mode 'index' expects the neuron_selection parameter.")  # base.py:489-492
        return self._impl.analyze(X)
