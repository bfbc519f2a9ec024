"""MI355X-native LRP engine for attention image-captioning models.

Drop-in for the explanation hot path of SunJiamei/LRP-ImageCaptioning
(models/explainers.py + the vendored iNNvestigate LRPSequentialPresetA):
the per-token LRP backward pass through the adaptive-attention / grid-TD
LSTM decoder and on through the VGG16 encoder, as hand-written HIP kernels
for gfx950 behind a C ABI (include/lrp_hip.h).  See DESIGN.md.
"""
__version__ = "0.1.0"

from . import synthetic  # noqa: F401
