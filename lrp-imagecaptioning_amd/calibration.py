"""Per-model choice of the layers that may take the two-MFMA form of the opt-in fast mode (`lrp_set_fast_layers`).

The fast mode (`LRP_PREC_F16X2`, include/lrp_hip.h) multiplies the relevance with ONE fp16 per weight in the layers it
covers.  That rounding is the same for every token, so its effect on a heat-map depends on the model: it averages out over
the thousands of comparable products of a dense Gaussian kernel and does not when a sum has a few dominant products (sparse,
heavy-tailed, trained-like kernels: tests/test_gpu_stress_parity.py).  Whether a given network tolerates it is therefore
MEASURED here, on the caller's own weights and images, instead of assumed:

  reference   the exact-fp32 mode (the reference's arithmetic: TF float32, RR:274-322) on the same images and the same
              relevance at the top of the encoder — the CNN half in isolation, because that is the half the mode changes (the
              decoder is fp32 / fp64 in every mode);
  floor       fp16 pairs with three MFMAs in every layer (mask 0);
  per layer   the two-term form in that layer alone;
  the mask    layers are added in order of increasing individual error for as long as the error of the UNION, measured again
              after every addition, stays below tolerance / margin.

The result is a statement about the calibration images: a measured margin (default 10x below the 1e-4 bar), not a bound.
`LRPEngine.set_precision("f16x2")` + `set_fast_layers(mask)` then run the chosen mix; mask 0 ("fp16 pairs, three MFMAs
everywhere") is what a model that tolerates nothing ends up with, and is at least as exact as the default split-bf16 walk.
"""
import numpy as np
import torch


def _rel_l1(a, b):
    """max over heat-maps of sum|a - b| / sum|b| (BASELINE.json's metric), on the device"""
    n = a.shape[0]
    num = (a.double() - b.double()).abs().reshape(n, -1).sum(1)
    den = b.double().abs().reshape(n, -1).sum(1)
    return float((num / den.clamp_min(1e-300)).max())


def default_relevances(feat, seed=0):
    """Three relevance maps per image at the top of the encoder: dense N(0,1) x features, one-hot at the largest feature, the 20
    largest features — the shapes the decoder's LRP produces (diffuse early words, concentrated late ones)."""
    B = feat.shape[0]
    g = torch.Generator(device="cpu").manual_seed(seed)
    out, idx = [], []
    for b in range(B):
        f = feat[b]
        flat = f.reshape(-1)
        dense = torch.randn(f.shape, generator=g).to(f.device) * f
        order = torch.argsort(flat, descending=True)
        one = torch.zeros_like(flat)
        one[order[0]] = flat[order[0]]
        top = torch.zeros_like(flat)
        top[order[:20]] = flat[order[:20]]
        out += [dense, one.reshape(f.shape), top.reshape(f.shape)]
        idx += [b, b, b]
    return torch.stack(out).float().contiguous(), idx


def calibrate_fast_mode(engine, images, relevances=None, img_idx=None, tolerance=1e-4, margin=10.0, apply=True):
    """Measure, on `engine`'s current weights, which conv layers may take the two-MFMA form.

    images      (B, H, W, 3) float32, B <= engine.max_images
    relevances  (n, L, D) float32 at the top of the encoder with img_idx (n,) — default: default_relevances of the features
    Returns dict(mask, layers, error, floor, per_layer={li: err}, budget, reference="fp32", candidates=[...]).
    With apply=True the engine is left in 'f16x2' mode with the chosen mask (the caller encodes its next batch as usual);
    otherwise mode and mask are restored."""
    prev_mode, prev_mask = engine.precision, getattr(engine, "fast_layers", -1)
    X = images if torch.is_tensor(images) else torch.as_tensor(np.ascontiguousarray(images, dtype=np.float32))
    X = X.to(engine.device)
    budget = tolerance / margin
    # validate what can be validated before the engine's mode is touched
    if X.dim() != 4 or X.shape[0] < 1 or X.shape[0] > engine.max_images:
        raise ValueError("images must be (B, H, W, 3) with 1 <= B <= max_images = %d" % engine.max_images)
    if relevances is not None:
        if img_idx is None or len(img_idx) != relevances.shape[0]:
            raise ValueError("relevances need one img_idx entry each")
        if relevances.shape[0] > engine.max_tokens:
            raise ValueError("%d calibration relevances > max_tokens = %d" % (relevances.shape[0], engine.max_tokens))
    elif 3 * X.shape[0] > engine.max_tokens:
        raise ValueError("%d calibration relevances > max_tokens = %d" % (3 * X.shape[0], engine.max_tokens))
    done = False
    try:
        # ---- reference: exact fp32, same images, same relevance
        engine.set_precision("fp32")
        engine.encode_images(X)
        feat = engine.get_features().clone()
        if relevances is None:
            relevances, img_idx = default_relevances(feat)
        R = relevances.to(engine.device).float().contiguous()
        idx = [int(i) for i in img_idx]
        ref = engine.cnn_explain(idx, R).clone()

        def run(mask):
            engine.set_precision("f16x2")
            engine.set_fast_layers(mask)
            engine.encode_images(X)
            return _rel_l1(engine.cnn_explain(idx, R), ref)

        n_conv = len(engine.cnn_cfg)
        floor = run(0)
        # candidates: the layers the built-in rule would take (two-term needs a sum long enough to make sense; the top block,
        # where the relevance is most concentrated, is tried last like every other layer — the measurement decides)
        cand = [li for li in range(1, n_conv) if 9 * engine.cnn_cfg[li][1] >= 576 and 9 * engine.cnn_cfg[li][2] >= 576]
        per_layer = {li: run(1 << li) for li in cand}
        mask, err = 0, floor
        if floor <= budget:
            for li in sorted(cand, key=lambda q: per_layer[q]):
                if per_layer[li] > budget:
                    break
                e = run(mask | (1 << li))
                if e <= budget:
                    mask, err = mask | (1 << li), e
        res = dict(mask=mask, layers=[engine.cnn_cfg[li][0] for li in range(n_conv) if (mask >> li) & 1], error=err, floor=floor,
                   per_layer={engine.cnn_cfg[li][0]: per_layer[li] for li in cand}, budget=budget, reference="fp32",
                   candidates=[engine.cnn_cfg[li][0] for li in cand], n_images=int(X.shape[0]), n_relevances=int(R.shape[0]))
        done = True
    finally:
        # every exit path leaves a defined state: the calibrated mix (apply, success) or the caller's mode and mask
        if done and apply:
            engine.set_precision("f16x2")
            engine.set_fast_layers(mask)
        else:
            engine.set_fast_layers(prev_mask)
            engine.set_precision(prev_mode)
    return res
