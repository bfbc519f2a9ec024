"""CPU tests of the host-side tail (postprocess / heat-map / LRP-inference scores)."""
import numpy as np
import pytest

from lrp_imagecaptioning_amd import postprocess as P
from lrp_imagecaptioning_amd.parallel import flatten_bundle, shard_range, unflatten_bundle


def test_postprocess_flips_channels_and_copies():
    X = np.arange(2 * 2 * 2 * 3, dtype=np.float32).reshape(2, 2, 2, 3)
    Y = P.postprocess(X, "BGRtoRGB", False)
    np.testing.assert_array_equal(Y, X[..., ::-1])
    Y[0, 0, 0, 0] = -1
    assert X[0, 0, 0, 2] == 2                   # the input is not modified


def test_project_and_gamma_known_answers():
    X = np.array([[-2.0, 0.0, 1.0]])
    np.testing.assert_allclose(P.project(X), [[0.0, 0.5, 0.75]])
    np.testing.assert_allclose(P.project(np.zeros((1, 3))), [[0.5, 0.5, 0.5]])     # absmax == 0 left alone
    g = P.gamma(np.array([-4.0, 0.0, 1.0, 4.0]), gamma=0.5)
    np.testing.assert_allclose(g, [-4.0, 0.0, 2.0, 4.0])


def test_heatmap_shape_and_sign_colors():
    R = np.zeros((1, 4, 4, 3), dtype=np.float32)
    R[0, 0, 0] = 1.0          # positive relevance -> red-ish
    R[0, 3, 3] = -1.0         # negative -> blue-ish
    H = P.heatmap(R)
    assert H.shape == (1, 4, 4, 3) and H.dtype == np.float32
    assert H[0, 0, 0, 0] > H[0, 0, 0, 2]
    assert H[0, 3, 3, 2] > H[0, 3, 3, 0]
    np.testing.assert_allclose(H[0, 1, 1], [1.0, 1.0, 1.0], atol=0.02)   # zero relevance = white centre of seismic


def test_lrp_inference_scores():
    R = np.zeros((1, 2, 2, 3), dtype=np.float32)
    R[0, 0, 0] = [3, 3, 3]
    R[0, 1, 1] = [-6, -6, -6]
    assert P.lrp_inference_score(R, "mean") == pytest.approx((0.5 - 1.0) / 4)
    assert P.lrp_inference_score(R, "pos_mean") == pytest.approx(0.5 / 4)
    hp = np.array([0.5, 0, 0, -1.0])
    assert P.lrp_inference_score(R, "quantile") == pytest.approx(np.quantile(hp, 0.9))
    assert P.lrp_inference_score(np.zeros((1, 2, 2, 3)), "mean") == 0.0
    with pytest.raises(NotImplementedError):
        P.lrp_inference_score(R, "median")


def test_shard_range_partitions_exactly():
    for n, w in [(256, 8), (10, 4), (3, 8), (0, 2)]:
        cuts = [shard_range(n, w, r) for r in range(w)]
        assert cuts[0][0] == 0 and cuts[-1][1] == n
        assert all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
        assert max(hi - lo for lo, hi in cuts) - min(hi - lo for lo, hi in cuts) <= 1


def test_bundle_roundtrip():
    rs = np.random.RandomState(0)
    shapes = {"b": (3,), "a_W": (2, 3, 4), "z": (1, 1)}
    w = {k: rs.standard_normal(s).astype(np.float32) for k, s in shapes.items()}
    flat = flatten_bundle(w, shapes)
    assert flat.shape == (3 + 24 + 1,)
    back = unflatten_bundle(flat, shapes)
    for k in shapes:
        np.testing.assert_array_equal(back[k], w[k])
