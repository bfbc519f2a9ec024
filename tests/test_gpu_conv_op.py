"""-m gpu: the fp32-MFMA implicit-GEMM conv kernel (lrp_op_conv through the C ABI)
against float64 torch-CPU convolutions.  Covers every tile configuration
(N>64, N==64, N<=32), ragged M tails, Cin not a multiple of 32, 1x1 mode and
both LRP epilogues (gate multiply, gate multiply through a 2x2 pool)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l1
from gpu_util import report

pytestmark = pytest.mark.gpu


def _ref_conv(x, w, b, taps, relu):
    xt = torch.as_tensor(x, dtype=torch.float64).permute(0, 3, 1, 2)
    wt = torch.as_tensor(w, dtype=torch.float64).permute(3, 2, 0, 1)
    y = F.conv2d(xt, wt, torch.as_tensor(b, dtype=torch.float64), padding=1 if taps == 9 else 0)
    if relu:
        y = F.relu(y)
    return y.permute(0, 2, 3, 1).numpy()


def _ref_convT(s, w, taps):
    st = torch.as_tensor(s, dtype=torch.float64).permute(0, 3, 1, 2)
    wt = torch.as_tensor(w, dtype=torch.float64).permute(3, 2, 0, 1)      # (Cout,Cin,kh,kw)
    y = F.conv_transpose2d(st, wt, padding=1 if taps == 9 else 0)
    return y.permute(0, 2, 3, 1).numpy()


FWD_CASES = [  # NB, H, W, Cin, Cout, taps
    (2, 8, 8, 32, 128, 9),      # big tile, exact
    (1, 14, 14, 64, 256, 9),    # big tile, two N tiles, ragged M (196)
    (3, 7, 5, 8, 64, 9),        # n64 tile, Cin < 32, odd sizes
    (2, 6, 6, 16, 16, 9),       # n32 tile, tiny
    (1, 9, 9, 36, 40, 9),       # Cin, Cout not multiples of 32
    (1, 1, 300, 64, 96, 1),     # 1x1 mode (dense layer), ragged M
    (2, 16, 16, 128, 192, 9),   # N=192: one full + one half big tile
]


@pytest.mark.parametrize("case", FWD_CASES)
@pytest.mark.parametrize("mode", [0, 1])
def test_conv_forward(case, mode):
    from lrp_imagecaptioning_amd.engine import op_conv
    NB, H, W, Cin, Cout, taps = case
    rs = np.random.RandomState(hash(case) % 1000)
    k = 3 if taps == 9 else 1
    x = rs.standard_normal((NB, H, W, Cin)).astype(np.float32)
    w = (rs.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(np.float32)
    b = rs.standard_normal(Cout).astype(np.float32)
    out = op_conv(torch.as_tensor(x).cuda(), w, b, None, mode, taps).cpu().numpy()
    ref = _ref_conv(x, w, b, taps, relu=(mode == 0))
    err = rel_l1(out, ref)
    report("conv_fwd", case=list(case), mode=mode, rel_l1=err)
    assert out.shape == ref.shape
    assert err < 2e-6, err


BWD_CASES = [  # NB, H, W, Cin(out channels of the LRP step), Cout(channels of S)
    (2, 8, 8, 128, 64),
    (1, 14, 14, 256, 128),
    (3, 6, 6, 64, 32),
    (2, 6, 6, 16, 24),
    (5, 4, 4, 8, 8),
]


@pytest.mark.parametrize("case", BWD_CASES)
@pytest.mark.parametrize("mode", [2, 3])
def test_conv_lrp_backward(case, mode):
    """out = convT(S, w) * gate   (mode 3: through a 2x2 max-pool, gate at 2x resolution)."""
    from lrp_imagecaptioning_amd.engine import op_conv
    NB, H, W, Cin, Cout = case
    rs = np.random.RandomState(sum(case))
    s = rs.standard_normal((NB, H, W, Cout)).astype(np.float32)
    w = np.abs(rs.standard_normal((3, 3, Cin, Cout)) / np.sqrt(9 * Cin)).astype(np.float32)
    up = 2 if mode == 3 else 1
    gate = rs.uniform(0, 1, size=(NB, up * H, up * W, Cin)).astype(np.float32)
    out = op_conv(torch.as_tensor(s).cuda(), w, None, torch.as_tensor(gate).cuda(), mode, 9).cpu().numpy()
    c = _ref_convT(s, w, 9)
    if mode == 3:
        c = c.repeat(2, axis=1).repeat(2, axis=2)
    ref = c * gate
    err = rel_l1(out, ref)
    report("conv_bwd", case=list(case), mode=mode, rel_l1=err)
    assert out.shape == ref.shape
    assert err < 2e-6, err


def test_identity_weight_asymmetric():
    """A = I check with an asymmetric operand (catches a transposed C/D map)."""
    from lrp_imagecaptioning_amd.engine import op_conv
    C = 64
    x = np.arange(2 * 4 * 4 * C, dtype=np.float32).reshape(2, 4, 4, C) / 100.0
    w = np.zeros((3, 3, C, C), dtype=np.float32)
    w[1, 1] = np.eye(C)
    w[1, 1, 3, 5] = 2.0                     # asymmetric: out[..,5] += 2*x[..,3]
    out = op_conv(torch.as_tensor(x).cuda(), w, np.zeros(C, np.float32), None, 1, 9).cpu().numpy()
    ref = x.copy()
    ref[..., 5] += 2.0 * x[..., 3]
    np.testing.assert_allclose(out, ref, rtol=1e-6, atol=1e-6)
