"""Seeded synthetic weights / inputs for the LRP hot path.

No trained checkpoints or datasets can be fetched, so parity tests, the golden
generator (tests/golden/make_golden.py) and bench.py all draw weights from the
same seeded generators.  The ORDER of the RandomState draws is part of the
fixture contract: full-size golden files store only the seed.

Weight naming follows the Keras layers the reference reads them from
(models/explainers.py:264-278 adaptive, :1000-1019 grid-TD); every matrix is
(input_dim, output_dim), i.e. y = x @ W + b, LSTM gate order i,f,g,o.
"""
import numpy as np


def _n(rs, *shape, fan_in=None):
    s = 1.0 if fan_in is None else 1.0 / np.sqrt(fan_in)
    return (rs.standard_normal(shape) * s).astype(np.float32)


def adaptive_weights(rs, L, D, H, E, V):
    """Adaptive-attention decoder (model.py:415-604)."""
    assert E == H
    w = {}
    w["image_features_W"] = _n(rs, D, H, fan_in=D)
    w["image_features_b"] = _n(rs, H) * 0.1
    w["global_W"] = _n(rs, D, E, fan_in=D)
    w["global_b"] = _n(rs, E) * 0.1
    w["embedding"] = _n(rs, V, E) * 0.5
    w["lstm_Wi"] = _n(rs, 2 * E, 4 * H, fan_in=2 * E)
    w["lstm_Wh"] = _n(rs, H, 4 * H, fan_in=H)
    w["lstm_b"] = _n(rs, 4 * H) * 0.1
    w["Wv"] = _n(rs, H, H, fan_in=H)
    w["Wg"] = _n(rs, H, H, fan_in=H)
    w["V"] = _n(rs, H, 1, fan_in=H)
    w["Wx"] = _n(rs, 2 * E, H, fan_in=2 * E)
    w["Wh"] = _n(rs, H, H, fan_in=H)
    w["Ws"] = _n(rs, H, H, fan_in=H)
    w["output_W"] = _n(rs, H, V, fan_in=H)
    w["output_b"] = _n(rs, V) * 0.1
    return w


def gridtd_weights(rs, L, D, H, E, V):
    """Grid-TD (bottom-up/top-down) decoder (model.py:609-823)."""
    w = {}
    w["image_features_W"] = _n(rs, D, H, fan_in=D)
    w["image_features_b"] = _n(rs, H) * 0.1
    w["global_W"] = _n(rs, D, E, fan_in=D)
    w["global_b"] = _n(rs, E) * 0.1
    w["embedding"] = _n(rs, V, E) * 0.5
    w["td_Wi"] = _n(rs, H + 2 * E, 4 * H, fan_in=H + 2 * E)
    w["td_Wh"] = _n(rs, H, 4 * H, fan_in=H)
    w["td_b"] = _n(rs, 4 * H) * 0.1
    w["lang_Wi"] = _n(rs, 2 * H, 4 * H, fan_in=2 * H)
    w["lang_Wh"] = _n(rs, H, 4 * H, fan_in=H)
    w["lang_b"] = _n(rs, 4 * H) * 0.1
    w["W_va"] = _n(rs, H, H, fan_in=H)
    w["W_ha"] = _n(rs, H, H, fan_in=H)
    w["W_a"] = _n(rs, H, 1, fan_in=H)
    w["W_x"] = _n(rs, H + 2 * E, H, fan_in=H + 2 * E)
    w["W_h"] = _n(rs, H, H, fan_in=H)
    w["W_s"] = _n(rs, H, H, fan_in=H)
    w["output_W"] = _n(rs, H, V, fan_in=H)
    w["output_b"] = _n(rs, V) * 0.1
    return w


def decoder_inputs(rs, L, D, V, T):
    """CNN feature map (1, sqrt L, sqrt L, D) >= 0 and a caption of T word ids
    followed by EOS (tokenizer ids: EOS=1, SOS=2, words >= 3; model column = id-1,
    preprocessors.py:179-189)."""
    g = int(round(np.sqrt(L)))
    feat = np.maximum(rs.standard_normal((1, g, g, D)), 0).astype(np.float32)
    cap = [int(c) for c in rs.randint(3, V + 1, size=T)] + [1]
    return feat, cap


def decoder_case(kind, seed, L, D, H, V, T):
    """Everything a golden decoder case is built from, in fixture draw order."""
    rs = np.random.RandomState(seed)
    w = (adaptive_weights if kind == "adaptive" else gridtd_weights)(rs, L, D, H, H, V)
    feat, cap = decoder_inputs(rs, L, D, V, T)
    return w, feat, cap


# --------------------------------------------------------------------------- CNN
VGG16_CFG = [  # (name, C_in, C_out, pool_after)   keras.applications.vgg16 up to block5_conv3
    ("block1_conv1", 3, 64, False), ("block1_conv2", 64, 64, True),
    ("block2_conv1", 64, 128, False), ("block2_conv2", 128, 128, True),
    ("block3_conv1", 128, 256, False), ("block3_conv2", 256, 256, False), ("block3_conv3", 256, 256, True),
    ("block4_conv1", 256, 512, False), ("block4_conv2", 512, 512, False), ("block4_conv3", 512, 512, True),
    ("block5_conv1", 512, 512, False), ("block5_conv2", 512, 512, False), ("block5_conv3", 512, 512, False),
]


VGG19_CFG = [  # keras.applications.vgg19 up to block5_conv4 (config.py:37: the layer_name of the 'vgg19' encoder)
    ("block1_conv1", 3, 64, False), ("block1_conv2", 64, 64, True),
    ("block2_conv1", 64, 128, False), ("block2_conv2", 128, 128, True),
    ("block3_conv1", 128, 256, False), ("block3_conv2", 256, 256, False), ("block3_conv3", 256, 256, False),
    ("block3_conv4", 256, 256, True),
    ("block4_conv1", 256, 512, False), ("block4_conv2", 512, 512, False), ("block4_conv3", 512, 512, False),
    ("block4_conv4", 512, 512, True),
    ("block5_conv1", 512, 512, False), ("block5_conv2", 512, 512, False), ("block5_conv3", 512, 512, False),
    ("block5_conv4", 512, 512, False),
]


def vgg_weights(rs, cfg=VGG16_CFG, bias_std=0.05):
    """He-normal HWIO kernels (3,3,Cin,Cout) and mixed-sign biases (exercise the
    b+/b- split of relevance_rule.py:256-260).  ImageNet weights cannot be fetched."""
    w = {}
    for name, cin, cout, _ in cfg:
        w[name + "_W"] = (rs.standard_normal((3, 3, cin, cout)) * np.sqrt(2.0 / (9 * cin))).astype(np.float32)
        w[name + "_b"] = (rs.standard_normal((cout,)) * bias_std).astype(np.float32)
    return w


def vgg_weights_trained_like(rs, cfg=VGG16_CFG, density=0.05, sigma=1.5, active_frac=0.2, calib_image=None):
    """Kernels with the statistics of a TRAINED network instead of an initialisation: only `density` of the entries are
    non-zero and their magnitudes are heavy-tailed (He-normal x lognormal(0, sigma)), rescaled to the He variance
    2 / (9 C_in) (then per layer so that the post-ReLU rms on the calibration image is 1); biases are pushed negative channel by channel until only `active_frac` of a channel's post-ReLU
    activations on `calib_image` (1, H, W, 3) are non-zero (sparse activations, mostly-dead windows).  A sum of an
    alpha1beta0 layer then has a few dominant products instead of thousands of comparable ones — the case in which
    per-weight rounding errors do not average out (tests/test_gpu_stress_parity.py).  The calibration is a float32
    torch-CPU forward, layer by layer."""
    import torch
    import torch.nn.functional as F
    w = {}
    x = None if calib_image is None else torch.as_tensor(np.ascontiguousarray(calib_image)).permute(0, 3, 1, 2).contiguous()
    for name, cin, cout, pool in cfg:
        k = rs.standard_normal((3, 3, cin, cout))
        k *= (rs.uniform(size=k.shape) < density) * np.exp(sigma * rs.standard_normal(k.shape))
        k *= np.sqrt(2.0 / (9 * cin)) / max(float(k.std()), 1e-30)
        w[name + "_W"] = k.astype(np.float32)
        b = (rs.standard_normal((cout,)) * 0.05).astype(np.float32)
        if x is not None:
            z = F.conv2d(x, torch.as_tensor(w[name + "_W"]).permute(3, 2, 0, 1).contiguous(), None, padding=1)
            q = torch.quantile(z.permute(1, 0, 2, 3).reshape(cout, -1)[:, ::max(1, z[0, 0].numel() // 4096)], 1.0 - active_frac, dim=1)
            b = (-q).numpy().astype(np.float32)
            x = F.relu(z + torch.as_tensor(b).view(1, -1, 1, 1))
            g = 1.0 / max(float(x.pow(2).mean().sqrt()), 1e-30)      # keep the activations' scale: post-ReLU rms = 1
            w[name + "_W"] = (w[name + "_W"] * g).astype(np.float32)
            b = (b * g).astype(np.float32)
            x = x * g
            if pool:
                x = F.max_pool2d(x, 2, 2)
        w[name + "_b"] = b
    return w


def images(rs, B, H=224, W=224):
    """BGR 'caffe'-mode preprocessed images: U[0,255] - mean (preprocessors.py:43-44)."""
    mean = np.array([103.939, 116.779, 123.68], dtype=np.float32)
    return (rs.uniform(0, 255, size=(B, H, W, 3)).astype(np.float32) - mean).astype(np.float32)


def captions(rs, B, T, V):
    return [[int(c) for c in rs.randint(3, V + 1, size=T)] + [1] for _ in range(B)]


# --------------------------------------------------------------------------- ResNet (config 4)
RESNET101_STACKS = ((64, 3), (128, 4), (256, 23), (512, 3))


def resnet_conv_list(stacks=RESNET101_STACKS, stem=64):
    """(name, k, cin, cout, stride) of every conv in forward order — ResNet-v1 bottleneck
    (keras_applications.resnet_common: ResNet101 -> stack1 -> block1)."""
    out = [("conv1", 7, 3, stem, 2)]
    cin = stem
    for i, (f, n) in enumerate(stacks):
        for b in range(1, n + 1):
            p = "conv%d_block%d" % (i + 2, b)
            stride = (1 if i == 0 else 2) if b == 1 else 1
            if b == 1:
                out.append((p + "_0", 1, cin, 4 * f, stride))
            out.append((p + "_1", 1, cin, f, stride))
            out.append((p + "_2", 3, f, f, 1))
            out.append((p + "_3", 1, f, 4 * f, 1))
            cin = 4 * f
    return out


def resnet_weights(rs, stacks=RESNET101_STACKS, stem=64, bias_std=0.05):
    """He-normal kernels, mixed-sign conv biases, non-trivial BN statistics (ImageNet weights cannot be fetched)."""
    w = {}
    for name, k, cin, cout, _ in resnet_conv_list(stacks, stem):
        w[name + "_conv_W"] = (rs.standard_normal((k, k, cin, cout)) * np.sqrt(2.0 / (k * k * cin))).astype(np.float32)
        w[name + "_conv_b"] = (rs.standard_normal((cout,)) * bias_std).astype(np.float32)
        w[name + "_bn_gamma"] = rs.uniform(0.6, 1.4, size=cout).astype(np.float32)
        w[name + "_bn_beta"] = (rs.standard_normal(cout) * 0.2).astype(np.float32)
        w[name + "_bn_mean"] = (rs.standard_normal(cout) * 0.3).astype(np.float32)
        w[name + "_bn_var"] = rs.uniform(0.5, 1.5, size=cout).astype(np.float32)
    return w


# --------------------------------------------------------------------------- canned scores for the beam-search fixture
def canned_score_table(seed, V, n_images, positions=8):
    """Seeded tables of a deterministic stand-in for the captioner's next-word scores (tests/golden/beam_*.npz: the
    reference's own `_beam_search` is run on them, tests/test_beam.py runs ours)."""
    rs = np.random.RandomState(seed)
    return {"M": rs.standard_normal((V + 1, V)).astype(np.float32), "P": rs.standard_normal((positions, V)).astype(np.float32),
            "I": (rs.standard_normal((n_images, V)) * 0.7).astype(np.float32)}


def canned_next_word_scores(table, image, words):
    """Un-normalised float32 scores (V,) of the word after `words` (tokenizer ids so far, SOS excluded) for `image`."""
    last = words[-1] if len(words) else 0
    prev = words[-2] if len(words) > 1 else 0
    P = table["P"]
    sc = (table["M"][last] + np.float32(0.37) * table["M"][prev] + P[len(words) % len(P)] + table["I"][image]).astype(np.float32)
    sc[0] += np.float32(0.8 * len(words) - 2.5)          # EOS (tokenizer id 1 = column 0): unlikely early, likely late
    return sc
