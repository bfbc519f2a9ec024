"""Known-answer tests pinning the CNN-half oracle (oracle/cnn_lrp_ref.py).
The reference holds no numeric fixtures at this boundary ("parity unpinned",
SURVEY.md §8c): these hand-computed cases are what pins the restatement."""
import numpy as np
import pytest
import torch

from conftest import rel_l1
from lrp_imagecaptioning_amd.synthetic import vgg_weights
from oracle import cnn_lrp_ref as C

TINY_CFG = [("c1", 3, 8, False), ("c2", 8, 8, True), ("c3", 8, 16, False), ("c4", 16, 16, True),
            ("c5", 16, 16, False)]


def _conv_layer(W, b):
    return [("conv", np.asarray(W, dtype=np.float64), np.asarray(b, dtype=np.float64))]


def test_1x1_effective_single_pixel():
    """1x1 image, 3x3 'same' conv: only the centre tap sees data.
    R_in[c] = x+[c] w+[c] / (sum x+ w+ + sum x- w- + b) * R  (+ x- w- part)."""
    W = np.zeros((3, 3, 2, 1))
    W[1, 1, :, 0] = [2.0, -3.0]
    x = np.array([1.5, -0.5]).reshape(1, 1, 1, 2)          # x+ = [1.5, 0], x- = [0, -0.5]
    b = [0.25]
    R = np.array([4.0]).reshape(1, 1, 1, 1)
    Z = 1.5 * 2.0 + (-0.5) * (-3.0) + 0.25                 # x+w+ + x-w- + b
    want = np.array([1.5 * 2.0, (-0.5) * (-3.0)]) / Z * 4.0
    got = C.analyze(_conv_layer(W, b), x, R)
    np.testing.assert_allclose(got.ravel(), want, rtol=1e-12)
    # mixed signs that do NOT contribute: x+ with w-, x- with w+
    W2 = np.zeros((3, 3, 2, 1))
    W2[1, 1, :, 0] = [-2.0, 3.0]
    got2 = C.analyze(_conv_layer(W2, b), x, R)
    np.testing.assert_allclose(got2.ravel(), [0.0, 0.0], atol=0)    # Z = b only, nothing routed


def test_negative_bias_neuron_flips_sign():
    """All-negative-bias neuron with no activation: Z = b- < 0 -> S = R/Z < 0."""
    W = np.zeros((3, 3, 1, 1))
    W[1, 1, 0, 0] = 1.0
    x = np.array([2.0]).reshape(1, 1, 1, 1)
    got = C.analyze(_conv_layer(W, [-3.0]), x, np.array([1.0]).reshape(1, 1, 1, 1))
    np.testing.assert_allclose(got.ravel(), [2.0 * 1.0 / (2.0 - 3.0)], rtol=1e-12)


def test_safe_divide_exact_zero():
    """Z == 0 exactly -> R / 1e-7 (IL:456-458); no sign stabiliser otherwise."""
    a = torch.tensor([1.0, 1.0, 1.0], dtype=torch.float64)
    b = torch.tensor([0.0, 1e-9, -2.0], dtype=torch.float64)
    np.testing.assert_allclose(C.safe_divide(a, b).numpy(), [1e7, 1e9, -0.5], rtol=1e-12)
    W = np.zeros((3, 3, 1, 1))
    W[1, 1, 0, 0] = 1.0
    x = np.array([0.0]).reshape(1, 1, 1, 1)
    got = C.analyze(_conv_layer(W, [0.0]), x, np.array([5.0]).reshape(1, 1, 1, 1))
    np.testing.assert_allclose(got.ravel(), [0.0])          # x * (w * 5e7) = 0


def test_3x3_on_4x4_hand_loop():
    """3x3 'same' conv on a 4x4x2 input with mixed-sign weights / bias, against
    a plain python-loop evaluation of RR:274-322."""
    rs = np.random.RandomState(3)
    x = rs.standard_normal((1, 4, 4, 2))
    W = rs.standard_normal((3, 3, 2, 3))
    b = rs.standard_normal(3)
    R = rs.standard_normal((1, 4, 4, 3))
    xp, xn = x * (x >= 0), x * (x < 0)
    Wp, Wn = W * (W >= 0), W * (W < 0)
    Z = np.zeros((4, 4, 3))
    for h in range(4):
        for w in range(4):
            for kh in range(3):
                for kw in range(3):
                    hh, ww = h + kh - 1, w + kw - 1
                    if 0 <= hh < 4 and 0 <= ww < 4:
                        Z[h, w] += xp[0, hh, ww] @ Wp[kh, kw] + xn[0, hh, ww] @ Wn[kh, kw]
    Z += b
    S = R[0] / (Z + (Z == 0) * 1e-7)
    want = np.zeros((4, 4, 2))
    for h in range(4):
        for w in range(4):
            for kh in range(3):
                for kw in range(3):
                    hh, ww = h + kh - 1, w + kw - 1
                    if 0 <= hh < 4 and 0 <= ww < 4:
                        want[hh, ww] += xp[0, hh, ww] * (Wp[kh, kw] @ S[h, w]) + xn[0, hh, ww] * (Wn[kh, kw] @ S[h, w])
    got = C.analyze(_conv_layer(W, b), x, R)
    np.testing.assert_allclose(got[0], want, rtol=1e-10, atol=1e-12)
    # conservation identity: sum R_in = sum R * (Z - b) / Z
    np.testing.assert_allclose(got.sum(), (R[0] * (Z - b) / Z).sum(), rtol=1e-10)


def test_maxpool_routes_to_argmax():
    x = np.array([[1.0, 5.0], [3.0, 2.0]]).reshape(1, 2, 2, 1)
    got = C.analyze([("pool",)], x, np.array([7.0]).reshape(1, 1, 1, 1))
    np.testing.assert_array_equal(got.reshape(2, 2), [[0, 7.0], [0, 0]])


def test_relu_output_passes_relevance_through():
    """The fused ReLU does not alter R (KG:244-264): two stacked convs equal
    applying the rule twice with the post-ReLU activation as second input."""
    rs = np.random.RandomState(0)
    W1, b1 = rs.standard_normal((3, 3, 2, 4)), rs.standard_normal(4)
    W2, b2 = rs.standard_normal((3, 3, 4, 3)), rs.standard_normal(3)
    x = rs.standard_normal((1, 5, 5, 2))
    R = rs.standard_normal((1, 5, 5, 3))
    layers = _conv_layer(W1, b1) + _conv_layer(W2, b2)
    a1 = C.forward(layers[:1], x)
    assert (a1 >= 0).all()
    R1 = C.analyze(layers[1:], a1, R)
    want = C.analyze(layers[:1], x, R1)
    np.testing.assert_allclose(C.analyze(layers, x, R), want, rtol=1e-12)


def test_epsilon_dense_rule():
    x = np.array([[1.0, -2.0]])
    W = np.array([[3.0], [1.0]])
    got = C.epsilon_dense(x, W, np.array([[2.0]]), eps=0.01)
    z = 1.0 * 3 - 2.0 * 1
    np.testing.assert_allclose(got, [[3.0 / (z + 0.01) * 2, -2.0 / (z + 0.01) * 2]], rtol=1e-12)


def test_add_and_batchnorm_reverse():
    a, b = np.array([1.0, -2.0]), np.array([3.0, 2.0])
    ra, rb = C.add_reverse([a, b], np.array([8.0, 5.0]))
    np.testing.assert_allclose(ra, [2.0, -2.0 * 5.0 / 1e-7])
    np.testing.assert_allclose(rb, [6.0, 2.0 * 5.0 / 1e-7])
    x = np.array([[2.0]])
    got = C.batchnorm_reverse(x, [1.5], [0.5], [1.0], [4.0], 0.0, np.array([[3.0]]))
    y = (2.0 - 1.0) / 2.0 * 1.5 + 0.5
    np.testing.assert_allclose(got, [[2.0 * (y - 0.5) * 3.0 / ((2.0 - 1.0) * y + 1e-7)]], rtol=1e-12)


@pytest.mark.parametrize("seed", [0, 1])
def test_cached_algorithm_equals_literal_graph(seed):
    """The restructured 1-pass algorithm the HIP kernels implement (cache Z+,
    G = mask*a/Z, drop the zero x-/w- branch) equals the literal 5-pass
    iNNvestigate graph to float64 round-off."""
    rs = np.random.RandomState(seed)
    w = vgg_weights(rs, TINY_CFG, bias_std=0.3)
    layers = C.vgg_layers(w, TINY_CFG)
    X = rs.uniform(-120, 130, size=(2, 16, 16, 3))
    feat = C.forward(layers, X)
    assert feat.shape == (2, 4, 4, 16)
    R = rs.standard_normal(feat.shape) * feat        # decoder relevance is proportional to the features
    lit = C.analyze(layers, X, R)
    fast = C.analyze_cached(layers, X, R)
    assert lit.shape == X.shape and np.isfinite(lit).all()
    assert rel_l1(fast, lit) < 1e-11


def test_float32_literal_close_to_float64():
    """What the 1e-4 bar is measured against: the reference ran this graph in
    float32 (TF); float32 vs float64 evaluation of the same graph."""
    rs = np.random.RandomState(5)
    w = vgg_weights(rs, TINY_CFG)
    layers = C.vgg_layers(w, TINY_CFG)
    X = rs.uniform(-120, 130, size=(1, 16, 16, 3)).astype(np.float32)
    feat = C.forward(layers, X)
    R = (rs.standard_normal(feat.shape) * feat).astype(np.float32)
    r64 = C.analyze(layers, X, R, torch.float64)
    r32 = C.analyze(layers, X, R, torch.float32)
    assert rel_l1(r32, r64) < 1e-4


def test_avgpool_reverse_known_answer():
    """2x2 average pool: Z = mean = 2.5, S = R/Z = 4, R_in = x * S / 4 = x  (conserves: sum R_in = 10 = R);
    an all-zero window takes the SafeDivide branch and returns zeros."""
    x = np.array([[1.0, 2.0, 0.0, 0.0], [3.0, 4.0, 0.0, 0.0]]).reshape(1, 2, 4, 1)
    got = C.avgpool_reverse(x, 2, np.array([10.0, 7.0]).reshape(1, 1, 2, 1))
    np.testing.assert_allclose(got[0, :, :, 0], [[1.0, 2.0, 0.0, 0.0], [3.0, 4.0, 0.0, 0.0]], rtol=1e-12)


def test_gradient_walks_known_answers():
    """1x1-image, 3x3 conv acting as its centre tap: y = relu(w x + b).  dy/dx = w where the unit is active.
    Guided backprop additionally drops negative incoming values; input x gradient multiplies by x."""
    w = np.zeros((3, 3, 1, 2))
    w[1, 1, 0] = [2.0, -3.0]
    layers = [("conv", w, np.array([0.5, 10.0]))]
    X = np.array([1.0]).reshape(1, 1, 1, 1)                       # pre-activations: 2.5, 7.0 -> both active
    head = np.array([1.0, -1.0]).reshape(1, 1, 1, 2)
    assert C.gradient_analyze(layers, X, head, "gradient")[0, 0, 0, 0] == pytest.approx(2.0 * 1 + (-3.0) * (-1))
    assert C.gradient_analyze(layers, X, head, "guided_backprop")[0, 0, 0, 0] == pytest.approx(2.0)      # -1 clamped
    X2 = np.array([4.0]).reshape(1, 1, 1, 1)                      # second unit: -12 + 10 < 0 -> inactive
    assert C.gradient_analyze(layers, X2, head, "gradient")[0, 0, 0, 0] == pytest.approx(2.0)
    assert C.gradient_analyze(layers, X2, head, "input_x_gradient")[0, 0, 0, 0] == pytest.approx(8.0)


def test_gradient_walk_is_the_true_gradient():
    """The per-layer walk equals autograd through the whole network (fan-out free chain)."""
    import torch
    import torch.nn.functional as F
    rs = np.random.RandomState(3)
    cfg = [("a", 3, 8, True), ("b", 8, 8, False)]
    from lrp_imagecaptioning_amd.synthetic import vgg_weights
    wts = vgg_weights(rs, cfg, bias_std=0.3)
    layers = C.vgg_layers(wts, cfg)
    X = rs.standard_normal((2, 8, 8, 3))
    head = rs.standard_normal((2, 4, 4, 8))
    xt = torch.as_tensor(X).permute(0, 3, 1, 2).clone().requires_grad_(True)
    y = xt
    for L in layers:
        if L[0] == "conv":
            y = F.relu(F.conv2d(y, torch.as_tensor(L[1]).double().permute(3, 2, 0, 1), torch.as_tensor(L[2]).double(), padding=1))
        else:
            y = F.max_pool2d(y, 2, 2)
    (g,) = torch.autograd.grad(y, xt, grad_outputs=torch.as_tensor(head).permute(0, 3, 1, 2))
    np.testing.assert_allclose(C.gradient_analyze(layers, X, head, "gradient"), g.permute(0, 2, 3, 1).numpy(), rtol=1e-10, atol=1e-12)


def test_trained_like_generator_makes_what_the_stress_tests_rely_on():
    """synthetic.vgg_weights_trained_like (the weights of tests/test_gpu_stress_parity.py): sparse heavy-tailed kernels,
    ~80 % dead activations per conv, unit activation scale — checked on a small net with the float64 forward of the oracle."""
    from lrp_imagecaptioning_amd.synthetic import vgg_weights_trained_like
    cfg = [("c1", 3, 16, False), ("c2", 16, 16, True), ("c3", 16, 32, False), ("c4", 32, 32, False)]
    rs = np.random.RandomState(4)
    X = (rs.uniform(0, 255, size=(1, 32, 32, 3)) - 110).astype(np.float32)
    w = vgg_weights_trained_like(np.random.RandomState(5), cfg, density=0.1, sigma=1.5, active_frac=0.2, calib_image=X)
    for name, cin, cout, _ in cfg[1:]:
        k = w[name + "_W"]
        assert 0.03 < float((k != 0).mean()) < 0.2
        nz = np.abs(k[k != 0])
        assert nz.max() / np.median(nz) > 10                  # heavy tail
    layers = C.vgg_layers(w, cfg)
    feat, inputs = C.forward(layers, X, return_inputs=True)
    for i in range(1, len(layers)):
        if layers[i - 1][0] == "conv":
            a = inputs[i].numpy()
            assert 0.1 < float((a > 0).mean()) < 0.3          # calibrated in float32, evaluated in float64: ~20 % active
            assert 0.5 < float(np.sqrt((a ** 2).mean())) < 2.0
    assert np.isfinite(feat).all() and (feat > 0).any()
