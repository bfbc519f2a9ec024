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
