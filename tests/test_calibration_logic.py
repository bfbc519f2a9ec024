"""CPU: the selection logic of calibration.calibrate_fast_mode on a stand-in engine whose heat-map error is a known function
of the two-term mask — the greedy mix must stay under tolerance / margin, add layers in order of increasing individual error,
measure the UNION (errors need not add), fall back to mask 0 when nothing qualifies, and restore or apply the mode as asked."""
import numpy as np
import torch

from lrp_imagecaptioning_amd.calibration import calibrate_fast_mode, default_relevances

CFG = [("c0", 3, 64, False), ("c1", 64, 64, True), ("c2", 64, 128, False), ("c3", 128, 128, True), ("c4", 128, 8, False)]


class FakeEngine(object):
    """cnn_explain returns base * (1 + err(mask)) in f16x2 mode, base in fp32 mode; err adds per layer, plus an interaction term"""

    def __init__(self, per_layer, interaction=0.0, floor=1e-6):
        self.device = torch.device("cpu")
        self.cnn_cfg = CFG
        self.max_images, self.max_tokens = 4, 12
        self.precision, self.fast_layers = "bf16x3", -1
        self.per_layer, self.interaction, self.floor = per_layer, interaction, floor
        self.calls = []

    def set_precision(self, mode):
        self.precision = mode

    def set_fast_layers(self, mask):
        self.fast_layers = -1 if mask is None else mask

    def encode_images(self, X):
        self.n = X.shape[0]

    def get_features(self):
        g = torch.Generator().manual_seed(1)
        return torch.rand((self.n, 4, 8), generator=g) + 0.1

    def cnn_explain(self, idx, R):
        base = torch.ones((len(idx), 6, 6, 3), dtype=torch.float32)
        if self.precision == "fp32":
            return base
        m = max(self.fast_layers, 0)
        on = [li for li in range(len(CFG)) if (m >> li) & 1]
        err = self.floor + sum(self.per_layer.get(li, 1.0) for li in on) + self.interaction * max(len(on) - 1, 0)
        self.calls.append(m)
        return base * (1.0 + err)


def test_default_relevances_shapes():
    feat = torch.rand((2, 5, 7)) + 0.1
    R, idx = default_relevances(feat)
    assert R.shape == (6, 5, 7) and idx == [0, 0, 0, 1, 1, 1]
    assert int((R[1] != 0).sum()) == 1 and int((R[2] != 0).sum()) == 20       # one-hot, top-20
    assert float(R[1].max()) == float(feat[0].max())


def test_greedy_mix_respects_the_budget_and_the_order():
    # candidates: layers 1..3 (cin and cout >= 64); layer 4 (cout 8) and layer 0 are never tried
    eng = FakeEngine({1: 2e-6, 2: 6e-6, 3: 4e-6})
    res = calibrate_fast_mode(eng, np.zeros((2, 8, 8, 3), np.float32), tolerance=1e-4, margin=10.0)
    assert res["candidates"] == ["c1", "c2", "c3"] and res["budget"] == 1e-5
    # order of trial: c1 (2e-6), c3 (4e-6), c2 (6e-6): 1e-6 + 2e-6 + 4e-6 = 7e-6 fits, adding c2 would make 1.3e-5
    assert res["layers"] == ["c1", "c3"] and res["mask"] == (1 << 1) | (1 << 3)
    assert abs(res["error"] - 7e-6) < 1e-7 and abs(res["floor"] - 1e-6) < 1e-7
    assert eng.precision == "f16x2" and eng.fast_layers == res["mask"]         # apply=True leaves the chosen mix in place
    assert all((m & 1) == 0 and (m >> 4) == 0 for m in eng.calls)              # layer 0 / the narrow layer were never enabled


def test_union_is_measured_not_assumed():
    # individually fine, together not (an interaction term): only one layer is kept
    eng = FakeEngine({1: 3e-6, 2: 3e-6, 3: 3e-6}, interaction=5e-6)
    res = calibrate_fast_mode(eng, np.zeros((1, 8, 8, 3), np.float32))
    assert len(res["layers"]) == 1 and res["error"] <= res["budget"]


def test_nothing_qualifies_and_restore():
    eng = FakeEngine({1: 5e-5, 2: 2e-4, 3: 8e-5})
    eng.set_precision("bf16x3")
    res = calibrate_fast_mode(eng, np.zeros((1, 8, 8, 3), np.float32), apply=False)
    assert res["mask"] == 0 and res["layers"] == [] and res["error"] == res["floor"]
    assert eng.precision == "bf16x3" and eng.fast_layers == -1                 # apply=False restores mode and mask
    # a floor above the budget: no layer is even tried for the union
    eng2 = FakeEngine({1: 1e-7}, floor=5e-5)
    res2 = calibrate_fast_mode(eng2, np.zeros((1, 8, 8, 3), np.float32))
    assert res2["mask"] == 0 and res2["error"] == res2["floor"] > res2["budget"]


def test_an_exception_restores_the_callers_mode_and_mask():
    """inputs are validated before the engine is touched, and a failure inside the measurement loop restores the
    caller's mode and mask (try / finally) whether or not apply was asked for."""
    import pytest
    eng = FakeEngine({1: 2e-6})
    eng.precision, eng.fast_layers = "bf16x3", 6
    with pytest.raises(ValueError):                       # 3 relevances per image x 5 images > max_tokens = 12: refused up front
        calibrate_fast_mode(eng, np.zeros((4, 8, 8, 3), np.float32)[:, :, :, :], relevances=torch.zeros((13, 4, 8)), img_idx=[0] * 13)
    assert (eng.precision, eng.fast_layers, eng.calls) == ("bf16x3", 6, [])

    class Boom(FakeEngine):
        def cnn_explain(self, idx, R):
            if self.precision == "f16x2" and len(self.calls) >= 2:
                raise RuntimeError("LRP_ERR_HIP")
            return FakeEngine.cnn_explain(self, idx, R)
    for apply in (True, False):
        eng = Boom({1: 2e-6, 2: 3e-6, 3: 4e-6})
        eng.precision, eng.fast_layers = "fp32", 2
        with pytest.raises(RuntimeError):
            calibrate_fast_mode(eng, np.zeros((2, 8, 8, 3), np.float32), apply=apply)
        assert (eng.precision, eng.fast_layers) == ("fp32", 2)
