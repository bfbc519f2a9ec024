"""Relevance -> RGB heat-map tail of the path (host side, cheap), restating
innvestigate/examples/utils_imagenet.py:14-33 and innvestigate/utils/visualizations.py:36-125,
plus the score reductions of the LRP-inference layer (models/model.py:1642-1690)."""
import numpy as np


def postprocess(X, color_conversion=None, channels_first=False):
    """utils_imagenet.py:14-18 -> utils/__init__.py:123-139 for channels_last data."""
    X = np.array(X, copy=True)
    assert color_conversion in [None, "RGBtoBGR", "BGRtoRGB"]
    if color_conversion in ("RGBtoBGR", "BGRtoRGB"):
        X = X[:, :, :, ::-1]
    if channels_first:
        X = X.transpose(0, 3, 1, 2)
    return X


def project(X, output_range=(0, 1), absmax=None, input_is_postive_only=False):
    """visualizations.py:36-54 (per-sample abs-max normalisation)."""
    X = np.array(X, dtype=np.float64, copy=True)
    if absmax is None:
        absmax = np.max(np.abs(X), axis=tuple(range(1, X.ndim)))
    absmax = np.asarray(absmax, dtype=np.float64)
    mask = absmax != 0
    if mask.sum() > 0:
        X[mask] /= absmax[mask].reshape((-1,) + (1,) * (X.ndim - 1)) if absmax.ndim else absmax
    if not input_is_postive_only:
        X = (X + 1) / 2
    X = X.clip(0, 1)
    return output_range[0] + X * (output_range[1] - output_range[0])


def gamma(X, gamma=0.5, minamp=0, maxamp=None):
    """visualizations.py:87-125: sign-preserving gamma correction."""
    X = np.asarray(X)
    Y = np.zeros_like(X)
    X = X - minamp
    if maxamp is None:
        maxamp = np.abs(X).max()
    X = X / maxamp
    pos = X >= 0
    Y[pos] = X[pos] ** gamma
    Y[~pos] = -(-X[~pos]) ** gamma
    return Y * maxamp + minamp


def _seismic(idx):
    try:
        import matplotlib
        cmap = matplotlib.colormaps["seismic"] if hasattr(matplotlib, "colormaps") else None
        if cmap is None:
            from matplotlib import cm
            cmap = cm.get_cmap("seismic")
        return cmap(idx)[:, :3]
    except Exception:                                   # no matplotlib: piece-wise linear seismic
        t = np.asarray(idx, dtype=np.float64) / 255.0
        r = np.clip(np.where(t < 0.5, 2 * t, np.where(t < 0.75, 1.0, 1.0 - 2 * (t - 0.75))), 0, 1)
        g = np.clip(np.where(t < 0.5, 2 * t, 2 * (1 - t)), 0, 1)
        b = np.clip(np.where(t < 0.25, 0.3 + 2.8 * t, np.where(t < 0.5, 1.0, 2 * (1 - t))), 0, 1)
        return np.stack([r, g, b], axis=-1)


def ivis_heatmap(X, reduce_axis=-1):
    """visualizations.py:57-80 with cmap 'seismic', reduce_op 'sum'."""
    X = np.asarray(X)
    shape = list(X.shape)
    tmp = X.sum(axis=reduce_axis)
    tmp = project(tmp, output_range=(0, 255)).astype(np.int64)
    rgb = _seismic(tmp.flatten())
    shape[reduce_axis] = 3
    return rgb.reshape(shape).astype(np.float32)


def heatmap(X):
    """utils_imagenet.py:31-33."""
    return ivis_heatmap(gamma(X, minamp=0, gamma=0.95))


def lrp_inference_score(relevance, mode, color_conversion="BGRtoRGB"):
    """models/model.py:1675-1686: one scalar per heat-map.  relevance (1,H,W,3) or (H,W,3)."""
    r = np.asarray(relevance, dtype=np.float32)
    if r.ndim == 3:
        r = r[None]
    hp = postprocess(r, color_conversion, False)
    hp = np.mean(hp, axis=-1)[0]
    m = np.max(np.abs(hp))
    hp = np.zeros(hp.shape) if m == 0 else 1.0 * hp / m
    if mode == "mean":
        return float(np.mean(hp))
    if mode == "pos_mean":
        return float(np.mean(np.maximum(hp, 0)))
    if mode == "quantile":
        return float(np.quantile(hp, [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9])[8])
    raise NotImplementedError("the lrp inference mode is not available")
