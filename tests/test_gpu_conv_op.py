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


# ---- split-bf16 path, incl. the halo-resident 3x3 variant (LRP_CONV_HALO: 0 never, 1 auto, 2 always)
SPLIT_CASES = [  # NB, H, W, Cin, Cout   (forward: Cin -> Cout; backward: S has Cout channels, out has Cin)
    (3, 14, 14, 64, 128),     # tw = 14: tiles span image boundaries in the stack
    (2, 28, 28, 40, 64),      # tw = 14 on W = 28 (BM = 128), Cin not a multiple of 32
    (1, 56, 56, 8, 128),      # two column tiles per row
    (2, 7, 5, 72, 64),        # W < every candidate: ragged columns, tiny image
    (1, 9, 33, 16, 128),      # W = 33 = 3 x 11
    (2, 16, 16, 64, 64),      # power-of-two width (the halo pitch equals tw + 2 only for 14 / 30)
    (33, 56, 56, 8, 256),     # 8-wave 256 x 256 tile (>= 400 blocks), tw = 28
    (140, 14, 14, 16, 256),   # 8-wave tile on 14 x 14 images: 18 stack rows per tile, ~1.3 images
    (3, 28, 28, 64, 64),      # backward: N = 64, two channel chunks -> weights-in-registers kernel (one group)
    (2, 14, 28, 64, 128),     # backward: N = 64, four chunks = two groups of the resident image
    (2, 14, 14, 56, 96),      # backward: N = 56 (< 64), three chunks: the last group holds one chunk
]


@pytest.fixture(params=["0", "1", "2"])
def halo_mode(request):
    from lrp_imagecaptioning_amd.engine import switches
    with switches(LRP_CONV_HALO=request.param):
        yield request.param


@pytest.mark.parametrize("case", SPLIT_CASES)
def test_split_bf16_forward(case, halo_mode):
    from lrp_imagecaptioning_amd.engine import op_conv
    NB, H, W, Cin, Cout = case
    rs = np.random.RandomState(sum(case))
    x = rs.standard_normal((NB, H, W, Cin)).astype(np.float32)
    w = (rs.standard_normal((3, 3, Cin, Cout)) / np.sqrt(9 * Cin)).astype(np.float32)
    b = rs.standard_normal(Cout).astype(np.float32)
    out = op_conv(torch.as_tensor(x).cuda(), w, b, None, 1, 9, split_bf16=True).cpu().numpy()
    err = rel_l1(out, _ref_conv(x, w, b, 9, relu=False))
    report("conv_split_fwd", case=list(case), halo=halo_mode, rel_l1=err)
    assert err < 2e-5, err


@pytest.mark.parametrize("case", SPLIT_CASES)
@pytest.mark.parametrize("mode", [2, 3])
def test_split_bf16_lrp_backward(case, mode, halo_mode):
    from lrp_imagecaptioning_amd.engine import op_conv
    NB, H, W, Cin, Cout = case
    if mode == 3 and NB * H * W > 50000:
        NB = max(1, NB // 4)          # the 2x-resolution gate of the pool mode is 4x the pixels
    rs = np.random.RandomState(sum(case) + mode)
    s = rs.standard_normal((NB, H, W, Cout)).astype(np.float32)
    w = np.abs(rs.standard_normal((3, 3, Cin, Cout)) / np.sqrt(9 * Cin)).astype(np.float32)
    up = 2 if mode == 3 else 1
    gate = rs.uniform(0, 1, size=(NB, up * H, up * W, Cin)).astype(np.float32)
    out = op_conv(torch.as_tensor(s).cuda(), w, None, torch.as_tensor(gate).cuda(), mode, 9, split_bf16=True).cpu().numpy()
    c = _ref_convT(s, w, 9)
    if mode == 3:
        c = c.repeat(2, axis=1).repeat(2, axis=2)
    err = rel_l1(out, c * gate)
    report("conv_split_bwd", case=list(case), mode=mode, halo=halo_mode, rel_l1=err)
    assert err < 2e-5, err


# ---- small grids take 64 x 64 tiles with a 4-stage k pipeline (LRP_CONV_SMALL, conv_igemm.h): every output element sees the
# same chain of MFMAs in the same k order as with the large tiles, so the two must agree BIT FOR BIT — that is what keeps a
# single image's heat-maps identical to the same image explained inside a batch of 32
SMALL_CASES = [  # NB, H, W, Cin, Cout
    (1, 14, 14, 512, 512),    # block5 of one image: 2 x 4 large tiles, K = 4608
    (10, 14, 14, 256, 128),   # ten words: 16 x 1 large tiles
    (1, 28, 28, 72, 64),      # N = 64 tile family, Cin not a multiple of 32
    (2, 7, 5, 40, 192),       # ragged M, N = 192
    (10, 28, 28, 256, 256),   # 62 x 2 large tiles (forward) / 62 x 2 (backward): between 128 and 256 -> the 128 x 64 tiles
]


@pytest.mark.parametrize("case", SMALL_CASES)
@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("mode", [1, 2, 3])
def test_small_tiles_are_bit_identical_to_large_tiles(case, split, mode):
    from lrp_imagecaptioning_amd.engine import op_conv
    NB, H, W, Cin, Cout = case
    rs = np.random.RandomState(sum(case) + mode)
    if mode == 1:
        x = rs.standard_normal((NB, H, W, Cin)).astype(np.float32)
        w = (rs.standard_normal((3, 3, Cin, Cout)) / np.sqrt(9 * Cin)).astype(np.float32)
        args = (torch.as_tensor(x).cuda(), w, rs.standard_normal(Cout).astype(np.float32), None, 1, 9)
    else:
        s = rs.standard_normal((NB, H, W, Cout)).astype(np.float32)
        w = np.abs(rs.standard_normal((3, 3, Cin, Cout)) / np.sqrt(9 * Cin)).astype(np.float32)
        up = 2 if mode == 3 else 1
        gate = rs.uniform(0, 1, size=(NB, up * H, up * W, Cin)).astype(np.float32)
        args = (torch.as_tensor(s).cuda(), w, None, torch.as_tensor(gate).cuda(), mode, 9)
    outs = {}
    from lrp_imagecaptioning_amd.engine import switches
    for small in ("1", "0"):
        with switches(LRP_CONV_SMALL=small, LRP_CONV_MID=small):   # the 128 x 64 tiles of the grids in between follow the same switch here
            outs[small] = op_conv(*args, split_bf16=split).clone()
    assert torch.equal(outs["1"], outs["0"]), float((outs["1"] - outs["0"]).abs().max())
