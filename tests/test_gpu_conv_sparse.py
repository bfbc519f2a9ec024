"""-m gpu: the 2:4-sparse consumer of a pooled boundary (csrc/conv_sparse.h, lrp_op_conv_pool_sparse) against
  * the dense split-bf16 kernel on the EXPANDED tensor (lrp_op_conv mode 2 | LRP_CONV_SPLIT_BF16): the same three-MFMA products,
    another summation order -> 1e-6;
  * a float64 torch reference of what both compute: conv_transpose of the expanded relevance with w+, times the gate
    (AlphaBetaRule RR:274-322 behind MaxPooling2D's gradient routing RA:470-480 -> IL:138-157) -> split-bf16's 2e-5.
Shapes: every parity class, tiles that span tokens, ragged last tiles, image borders, one and two column tiles, N = 256 / 512."""
import numpy as np
import pytest
import torch

from gpu_util import report

pytestmark = pytest.mark.gpu

CASES = [  # NB, Hp, Wp, Cin (output columns N), Cout (K side)
    (3, 14, 14, 256, 32),     # block4_conv3's geometry, tiles span tokens (18 window rows per tile, 14 per token)
    (2, 28, 28, 256, 48),     # two column tiles, three chunks
    (5, 7, 5, 256, 16),       # ragged: Wp < 14, Hp odd
    (1, 3, 17, 512, 64),      # two N tiles, ragged second column tile
    (40, 14, 14, 512, 512),   # block4_conv3 itself at 40 words
]


def _expand(sc, pos):
    NB, Hp, Wp, C = sc.shape
    S = torch.zeros((NB, 2 * Hp, 2 * Wp, C), dtype=sc.dtype, device=sc.device)
    for p in range(4):
        S[:, (p >> 1)::2, (p & 1)::2, :] = torch.where(pos == p, sc, torch.zeros_like(sc))
    return S


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(map(str, c)))
def test_sparse_consumer_matches_dense_kernel_and_float64(case):
    from lrp_imagecaptioning_amd.engine import op_conv, op_conv_pool_sparse
    NB, Hp, Wp, Cin, Cout = case
    rs = np.random.RandomState(sum(case))
    sc = torch.as_tensor(rs.standard_normal((NB, Hp, Wp, Cout)).astype(np.float32)).cuda()
    pos = torch.as_tensor(rs.randint(0, 4, size=(NB, Hp, Wp, Cout)).astype(np.uint8)).cuda()
    w = np.abs(rs.standard_normal((3, 3, Cin, Cout)) / np.sqrt(9 * Cout)).astype(np.float32)     # w+ >= 0 like the walk's
    w[rs.uniform(size=w.shape) < 0.3] = 0.0
    gate = torch.as_tensor(rs.uniform(0, 1, size=(NB, 2 * Hp, 2 * Wp, Cin)).astype(np.float32)).cuda()
    got = op_conv_pool_sparse(sc, pos, w, gate)
    assert bool(torch.isfinite(got).all())
    S = _expand(sc, pos)
    dense = op_conv(S, w, None, gate, 2, 9, split_bf16=True)
    den = float(dense.double().abs().sum())
    e_dense = float((got.double() - dense.double()).abs().sum()) / den
    # float64: out = gate * conv_transpose(S, w)  (lrp_op_conv's backward modes: the transposed 3x3 'same' conv of HWIO w)
    if NB * Hp * Wp * Cin * Cout <= 3 * 14 * 14 * 256 * 64 * 4:
        wt = torch.as_tensor(w).double().cuda().permute(3, 2, 0, 1).contiguous()                     # (Cout, Cin, 3, 3)
        ref = torch.nn.functional.conv_transpose2d(S.double().permute(0, 3, 1, 2), wt, padding=1).permute(0, 2, 3, 1) * gate.double()
        e64 = float((got.double() - ref).abs().sum() / ref.abs().sum())
        e64d = float((dense.double() - ref).abs().sum() / ref.abs().sum())
    else:
        e64 = e64d = float("nan")
    report("conv_sparse_%s" % "x".join(map(str, case)), vs_dense_kernel=e_dense, vs_float64=e64, dense_vs_float64=e64d)
    assert e_dense < 1e-6, e_dense
    if e64 == e64:
        assert e64 < 2e-5, (e64, e64d)
