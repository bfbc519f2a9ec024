"""-m gpu: the per-model choice of the fast mode's two-MFMA layers (lrp_set_fast_layers, calibration.py).

On dense He-normal kernels every candidate layer below the top block qualifies with a 10x margin under the 1e-4 bar; on
trained-like kernels (sparse, heavy-tailed, 80 % dead activations) the built-in rule leaves the bar, and the calibration ends
with a mix whose MEASURED error keeps the margin — by taking the three-MFMA product wherever the two-term form costs too
much.  Reference of the measurement: the exact-fp32 mode on the same images and relevances (the reference's arithmetic)."""
import numpy as np
import pytest
import torch

from gpu_util import report
from lrp_imagecaptioning_amd.synthetic import VGG16_CFG, adaptive_weights, images, vgg_weights, vgg_weights_trained_like

pytestmark = pytest.mark.gpu


def _engine(w, n_img):
    from lrp_imagecaptioning_amd.engine import LRPEngine
    eng = LRPEngine(decoder="adaptive", cnn_cfg=VGG16_CFG, img_hw=(224, 224), L=196, D=512, H=32, E=32, V=50,
                    max_images=n_img, max_tokens=3 * n_img, max_caption_len=4)
    w = dict(w)
    w.update(adaptive_weights(np.random.RandomState(3), 196, 512, 32, 32, 50))
    eng.set_weights(w)
    return eng


def _check_applied(eng, res, X):
    """the engine is left in f16x2 mode with the chosen mask, and what it then computes is what was measured"""
    from lrp_imagecaptioning_amd.calibration import _rel_l1, default_relevances
    assert eng.precision == "f16x2" and eng.fast_layers == res["mask"]
    Xd = torch.as_tensor(X).cuda()
    eng.encode_images(Xd)
    R, idx = default_relevances(eng.get_features())
    got = eng.cnn_explain(idx, R).clone()
    eng.set_precision("fp32")
    eng.encode_images(Xd)
    R32, _ = default_relevances(eng.get_features())
    ref = eng.cnn_explain(idx, R32)
    # (the relevances are rebuilt from each mode's own features here: 7e-7 apart, so this is a slightly different experiment
    #  than the calibration's — same order of magnitude is what is asserted)
    assert _rel_l1(got, ref) < 5 * max(res["budget"], res["error"])


def test_dense_gaussian_kernels_qualify_below_the_top_block():
    from lrp_imagecaptioning_amd.calibration import calibrate_fast_mode
    rs = np.random.RandomState(0)
    w = vgg_weights(rs, VGG16_CFG)
    X = images(np.random.RandomState(1), 2)
    eng = _engine(w, 2)
    res = calibrate_fast_mode(eng, X)
    report("calibration_he_normal", **{k: v for k, v in res.items() if k != "per_layer"}, **{"pl_" + k: v for k, v in res["per_layer"].items()})
    assert res["floor"] < 1e-5 and res["error"] <= res["budget"] == pytest.approx(1e-5)
    rule = ["block1_conv2", "block2_conv1", "block2_conv2", "block3_conv1", "block3_conv2", "block3_conv3", "block4_conv1",
            "block4_conv2", "block4_conv3"]
    assert len(res["layers"]) >= 6 and set(res["layers"]) & set(rule)     # most of what the rule takes, by measurement
    _check_applied(eng, res, X)


def test_trained_like_kernels_get_a_mix_that_keeps_the_margin():
    from lrp_imagecaptioning_amd.calibration import _rel_l1, calibrate_fast_mode, default_relevances
    X = images(np.random.RandomState(0), 1)
    w = vgg_weights_trained_like(np.random.RandomState(1), VGG16_CFG, 0.05, 1.5, 0.2, X)
    eng = _engine(w, 1)
    # what the built-in rule does on these weights (same measurement: fp32 mode as the reference)
    Xd = torch.as_tensor(X).cuda()
    eng.set_precision("fp32")
    eng.encode_images(Xd)
    R, idx = default_relevances(eng.get_features())
    ref = eng.cnn_explain(idx, R).clone()
    eng.set_precision("f16x2")
    eng.set_fast_layers(None)
    eng.encode_images(Xd)
    rule_err = _rel_l1(eng.cnn_explain(idx, R), ref)
    res = calibrate_fast_mode(eng, X)
    report("calibration_trained_like", rule_error=rule_err, **{k: v for k, v in res.items() if k != "per_layer"},
           **{"pl_" + k: v for k, v in res["per_layer"].items()})
    assert rule_err > 5e-5                                      # the reason the mode is not a default
    assert res["error"] <= max(res["budget"], res["floor"])     # the calibrated mix keeps the margin (or is the 3-MFMA floor)
    assert res["error"] < rule_err / 3
    _check_applied(eng, res, X)


def test_fast_layer_mask_api():
    rs = np.random.RandomState(0)
    cfg = [("c1", 3, 64, False), ("c2", 64, 64, True), ("c3", 64, 64, False)]
    from lrp_imagecaptioning_amd.engine import LRPEngine
    eng = LRPEngine(decoder="adaptive", cnn_cfg=cfg, img_hw=(16, 16), L=64, D=64, H=32, E=32, V=40, max_images=1, max_tokens=2,
                    max_caption_len=4)
    w = vgg_weights(rs, cfg)
    w.update(adaptive_weights(rs, 64, 64, 32, 32, 40))
    eng.set_weights(w)
    X = rs.uniform(-100, 100, size=(1, 16, 16, 3)).astype(np.float32)
    with pytest.raises(ValueError):
        eng.set_fast_layers(1)                                  # the image layer has no two-term form
    with pytest.raises(ValueError):
        eng.set_fast_layers(1 << 5)                             # beyond the configured convs
    eng.set_precision("f16x2")
    eng.encode_images(X)
    R = rs.standard_normal((1, 64, 64)).astype(np.float32)
    a = eng.cnn_explain([0], R).clone()
    eng.set_fast_layers([1, 2])                                 # a change in f16x2 mode drops the caches ...
    with pytest.raises(RuntimeError):
        eng.cnn_explain([0], R)
    eng.encode_images(X)                                        # ... until the images are encoded again
    b = eng.cnn_explain([0], R).clone()
    eng.set_fast_layers(0)
    eng.encode_images(X)
    c = eng.cnn_explain([0], R).clone()
    assert torch.isfinite(a).all() and torch.isfinite(b).all() and torch.isfinite(c).all()
    assert not torch.equal(b, c)                                # the mask reaches the kernels
    assert float((b - c).abs().sum() / c.abs().sum()) < 5e-3
    eng.set_precision("bf16x3")
    eng.set_fast_layers(None)                                   # no cache drop outside f16x2 mode
