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


# ---- Grad-CAM map of the Guided-Grad-CAM baselines (explainers.py:939-949 / :1643-1653)
def pyramid_expand(image, upscale=2, sigma=None):
    """skimage.transform.pyramid_expand(image, upscale, sigma, multichannel=False) restated with scipy (skimage is a
    third-party dependency the reference imports, explainers.py:14): bilinear `resize` to upscale x the shape
    (output pixel centres map to (o + 0.5) / upscale - 0.5, out-of-range samples mirrored, no anti-aliasing), then a
    Gaussian smoothing with `sigma` (scipy mode 'reflect').  Host-side: a (14,14) -> (224,224) map per word."""
    from scipy import ndimage as ndi
    image = np.asarray(image, dtype=np.float64)
    if sigma is None:
        sigma = 2 * upscale / 6.0
    out_shape = tuple(int(np.ceil(upscale * d)) for d in image.shape)
    coords = np.meshgrid(*[(np.arange(o) + 0.5) * (s / float(o)) - 0.5 for o, s in zip(out_shape, image.shape)], indexing="ij")
    resized = ndi.map_coordinates(image, coords, order=1, mode="mirror")
    return ndi.gaussian_filter(resized, sigma, mode="reflect")


def grad_cam(img_feature, grads, L, D, upscale=16):
    """explainers.py:939-949: channel weights = spatial mean of the feature gradients; cam = relu(expand(sum_c w_c F_c))
    / (max|cam| + 1e-6).  The reference hard-codes upscale=16 (14 -> 224) and sigma=20; `upscale` is the image / feature
    size ratio so that other encoder geometries work too."""
    g = int(np.sqrt(L))
    weights = np.mean(np.asarray(grads).reshape(g, g, D), axis=(0, 1))
    conv_output = np.asarray(img_feature).reshape(g, g, D)
    cam = np.tensordot(conv_output, weights, axes=([2], [0])).astype(np.float32)
    cam = pyramid_expand(cam, upscale=upscale, sigma=20)
    cam = np.maximum(cam, 0)
    return cam / (np.max(np.abs(cam)) + 1e-6)
