"""-m gpu: operator-level parity of the remaining preset rules
against the float64 oracle restatements (oracle/cnn_lrp_ref.py)."""
import numpy as np
import pytest
import torch

from conftest import rel_l1
from gpu_util import report
from oracle import cnn_lrp_ref as C

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,Din,Dout", [(5, 64, 32), (130, 512, 256), (3, 36, 20)])
def test_epsilon_dense_rule(N, Din, Dout):
    from lrp_imagecaptioning_amd.engine import op_epsilon_dense
    rs = np.random.RandomState(N)
    x = rs.standard_normal((N, Din)).astype(np.float32)
    W = (rs.standard_normal((Din, Dout)) / np.sqrt(Din)).astype(np.float32)
    R = rs.standard_normal((N, Dout)).astype(np.float32)
    out = op_epsilon_dense(torch.as_tensor(x).cuda(), W, torch.as_tensor(R).cuda(), 0.01).cpu().numpy()
    ref = C.epsilon_dense(x, W, R, 0.01)
    err = rel_l1(out, ref)
    report("rule_eps_dense", case=[N, Din, Dout], rel_l1=err)
    assert err < 1e-4


def test_batchnorm_reverse():
    from lrp_imagecaptioning_amd.engine import op_batchnorm_lrp
    rs = np.random.RandomState(0)
    Cc = 24
    x = rs.standard_normal((3, 7, 7, Cc)).astype(np.float32)
    g, b = rs.uniform(0.5, 1.5, Cc).astype(np.float32), rs.standard_normal(Cc).astype(np.float32)
    mu, var = rs.standard_normal(Cc).astype(np.float32), rs.uniform(0.5, 2, Cc).astype(np.float32)
    R = rs.standard_normal(x.shape).astype(np.float32)
    t = lambda a: torch.as_tensor(a).cuda()
    out = op_batchnorm_lrp(t(x), t(g), t(b), t(mu), t(var), 1.001e-5, t(R)).cpu().numpy()
    ref = C.batchnorm_reverse(x, g, b, mu, var, 1.001e-5, R)
    assert rel_l1(out, ref) < 1e-5


def test_add_reverse():
    from lrp_imagecaptioning_amd.engine import op_add_lrp
    rs = np.random.RandomState(1)
    a = rs.standard_normal((2, 5, 5, 16)).astype(np.float32)
    b = rs.standard_normal(a.shape).astype(np.float32)
    b.flat[3] = -a.flat[3]                      # exact-zero denominator -> SafeDivide branch
    R = rs.standard_normal(a.shape).astype(np.float32)
    t = lambda v: torch.as_tensor(v).cuda()
    Ra, Rb = op_add_lrp(t(a), t(b), t(R))
    ra, rb = C.add_reverse([a, b], R, dtype=torch.float32)
    assert rel_l1(Ra.cpu().numpy(), ra) < 1e-5 and rel_l1(Rb.cpu().numpy(), rb) < 1e-5
    assert np.isfinite(Ra.cpu().numpy()).all()


@pytest.mark.parametrize("mode", ["mean", "pos_mean", "quantile"])
def test_heatmap_scores_match_numpy(mode):
    """lrp_heatmap_scores against the host restatement of model.py:1675-1686 (postprocess.lrp_inference_score)."""
    from lrp_imagecaptioning_amd.engine import heatmap_scores
    from lrp_imagecaptioning_amd.postprocess import lrp_inference_score
    rs = np.random.RandomState(3)
    R = (rs.standard_normal((5, 37, 41, 3)) * rs.uniform(0.1, 10, size=(5, 1, 1, 1))).astype(np.float32)
    R[3] = 0.0                                        # all-zero map -> score 0
    R[4, :, :, :] = np.abs(R[4])                      # positive-only map
    got = heatmap_scores(torch.as_tensor(R).cuda(), mode).cpu().numpy()
    want = np.array([lrp_inference_score(R[i:i + 1], mode) for i in range(5)])
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=1e-9)
    with pytest.raises(NotImplementedError):
        heatmap_scores(torch.as_tensor(R).cuda(), "median")


def test_heatmap_scores_full_size_quantile_with_ties():
    from lrp_imagecaptioning_amd.engine import heatmap_scores
    from lrp_imagecaptioning_amd.postprocess import lrp_inference_score
    rs = np.random.RandomState(9)
    R = rs.standard_normal((2, 224, 224, 3)).astype(np.float32)
    R[1] = np.round(R[1] * 4) / 4                     # heavy ties around the selected order statistics
    got = heatmap_scores(torch.as_tensor(R).cuda(), "quantile").cpu().numpy()
    want = np.array([lrp_inference_score(R[i:i + 1], "quantile") for i in range(2)])
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=1e-9)


@pytest.mark.parametrize("shape,k", [((2, 14, 14, 32), 7), ((3, 8, 12, 8), 2), ((1, 7, 7, 2048), 7)])
def test_avgpool_reverse(shape, k):
    """AveragePoolingReverseLayer (RA:289-316); post-ReLU inputs plus one all-zero window (SafeDivide branch)."""
    from lrp_imagecaptioning_amd.engine import op_avgpool_lrp
    rs = np.random.RandomState(k)
    x = np.maximum(rs.standard_normal(shape), 0).astype(np.float32)
    x[0, :k, :k, 0] = 0.0
    R = rs.standard_normal((shape[0], shape[1] // k, shape[2] // k, shape[3])).astype(np.float32)
    out = op_avgpool_lrp(torch.as_tensor(x).cuda(), torch.as_tensor(R).cuda(), k).cpu().numpy()
    ref = C.avgpool_reverse(x, k, R)
    err = rel_l1(out, ref)
    report("rule_avgpool", case=list(shape) + [k], rel_l1=err)
    assert out.shape == x.shape and np.isfinite(out).all() and err < 1e-5
    assert (out[0, :k, :k, 0] == 0).all()


@pytest.mark.parametrize("h0,w0", [(375, 500), (224, 224), (100, 333), (640, 480), (128, 1000), (448, 2048)])   # incl. exact-tie scales
def test_preprocess_images_matches_pil(h0, w0):
    """lrp_preprocess_images against the host path (PIL nearest resize + caffe preprocess_input, harness.ImagePreprocessor)."""
    from PIL import Image
    from lrp_imagecaptioning_amd.engine import preprocess_images
    from lrp_imagecaptioning_amd.harness import VGG_BGR_MEAN
    rs = np.random.RandomState(h0)
    rgb = rs.randint(0, 256, size=(2, h0, w0, 3)).astype(np.uint8)
    out = preprocess_images(torch.as_tensor(rgb).cuda()).cpu().numpy()
    for i in range(2):
        ref = np.asarray(Image.fromarray(rgb[i]).resize((224, 224), Image.NEAREST), dtype=np.float32)[..., ::-1] - VGG_BGR_MEAN
        assert out[i].shape == ref.shape
        assert np.array_equal(out[i], ref.astype(np.float32))


def test_heatmap_render_matches_host():
    """lrp_heatmap_render against the host restatement of utils_imagenet.heatmap (postprocess.heatmap).  The colormap
    index is a truncation of a float32 pow chain, so single pixels may land one table entry off."""
    from lrp_imagecaptioning_amd.engine import heatmap_render
    from lrp_imagecaptioning_amd.postprocess import heatmap
    rs = np.random.RandomState(0)
    R = (rs.standard_normal((3, 56, 56, 3)) * rs.uniform(0.1, 3, size=(3, 1, 1, 1))).astype(np.float32)
    R[2] = 0.0        # all-zero map: numpy yields 0/0 = NaN and an undefined int cast; the device renders mid-scale
    got = heatmap_render(torch.as_tensor(R).cuda()).cpu().numpy()
    want = np.stack([heatmap(R[i:i + 1])[0] for i in range(3)])
    assert got.shape == want.shape == (3, 56, 56, 3)
    d = np.abs(got[:2] - want[:2]).max(axis=-1)
    assert (d > 0.02).mean() < 1e-3, (d > 0.02).mean()     # one LUT step of 'seismic' is <= 0.016 per channel
    assert np.isfinite(got).all()
    assert np.abs(got[2] - got[2, 0, 0]).max() == 0 and got[2, 0, 0].min() > 0.9      # index 127: white
