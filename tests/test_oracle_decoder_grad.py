"""Pin the gradient-baseline decoder oracle (oracle/decoder_grad_ref.py) to golden vectors produced by the
reference's own `_lstm_decoder_backward` (tests/golden/make_golden.py --only grad ran E:780-832 / E:1452-1532)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, rel_l1
from lrp_imagecaptioning_amd.synthetic import decoder_case
from oracle.decoder_grad_ref import AdaptiveGradOracle, GridTDGradOracle

SMALL = ["adaptive_grad_small_s0", "adaptive_grad_small_s1", "adaptive_grad_small_s2",
         "gridtd_grad_small_s0", "gridtd_grad_small_s1", "gridtd_grad_small_s2"]


def build(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    kind = str(g["kind"])
    L, D, H, E, V, T = [int(x) for x in g["dims"]]
    if "feat" in g.files:
        w = {k[2:]: g[k] for k in g.files if k.startswith("w_")}
        feat, cap = g["feat"], [int(c) for c in g["caption"]]
    else:
        w, feat, cap = decoder_case(kind, int(g["seed"]), L, D, H, V, T)
    o = (AdaptiveGradOracle if kind == "adaptive" else GridTDGradOracle)(w, L, D, H, E)
    o.forward(feat, cap)
    return g, kind, o


@pytest.mark.parametrize("name", SMALL)
def test_backward_matches_reference(name):
    g, kind, o = build(name)
    for n in (("ot_act", "gt_act") if kind == "adaptive" else ("o1t_act", "o2t_act", "g1t_act", "g2t_act")):
        np.testing.assert_allclose(getattr(o, n), g["state_" + n], rtol=2e-6, atol=1e-7, err_msg=n)
    for j, t in enumerate(g["tokens"]):
        d = o.backward(int(t))
        ref = g["d_feat"][j]
        assert d.shape == ref.shape and d.dtype == np.float32
        assert rel_l1(d, ref) < 2e-6, (t, rel_l1(d, ref))
        np.testing.assert_allclose(o.r_words, g["r_words_t%d" % t], rtol=2e-5, atol=1e-8)


@pytest.mark.parametrize("name", ["adaptive_grad_full_s0", "gridtd_grad_full_s0"])
def test_full_size(name):
    g, _, o = build(name)
    for j, t in enumerate(g["tokens"]):
        assert rel_l1(o.backward(int(t)), g["d_feat"][j]) < 5e-6
