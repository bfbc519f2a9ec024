"""ORACLE — test infrastructure only (never imported by the product path).

CPU restatement of `LRPSequentialPresetA(...).analyze([X, R])` for the ResNet-101 encoder
(keras_applications.resnet_common.ResNet101 cut at `conv5_block3_out`, config.py:41-45,
models/model.py:425-427): literal torch-autograd evaluation of the iNNvestigate reverse graph,
float64 by default.

The reference cannot run this configuration (explain_image.py:17-26 rejects the encoder,
models/explainers.py:29-30 hard-codes VGG layer names), keras/tensorflow are absent, and
keras_applications is not vendored, so **parity is unpinned by reference fixtures**: this file
restates the published architecture (ResNet v1 bottleneck: 7x7/2 stem with explicit zero
padding, 3x3/2 max-pool on a 1-padded map, stacks of [1x1(stride) -> 3x3 -> 1x1(4f)] with BN
after every conv, projection shortcut in the first block of each stack, Add, ReLU; BN epsilon
1.001e-5; every conv has a bias) and applies the rules of oracle/cnn_lrp_ref.py to it:

  Conv2D            -> Alpha1Beta0Rule (RR:274-322)         [kernel layer, RA:404-424]
  BatchNormalization-> BatchNormalizationReverseLayer (RA:197-257)
  Add               -> AddReverseLayer (RA:260-286)
  Activation(relu)  -> relevance passes through (RA:462-469)
  MaxPooling2D, ZeroPadding2D -> gradient routing (RA:470-480)
  a tensor consumed twice (block input: main path + shortcut) -> relevances are summed (KG:799-803)

It is pinned by (a) the known-answer tests of the shared rule functions (tests/test_oracle_cnn.py)
and (b) tests/test_oracle_resnet.py (conservation identities, hand-sized blocks).
"""
import numpy as np
import torch
import torch.nn.functional as F

from .cnn_lrp_ref import SAFE_EPS, safe_divide

BN_EPS = 1.001e-5


def resnet_spec(stacks=((64, 3), (128, 4), (256, 23), (512, 3)), stem=64):
    """List of (stack name, filters, n_blocks, stride of the first block) — ResNet-101 by default."""
    out = []
    for i, (f, n) in enumerate(stacks):
        out.append(("conv%d" % (i + 2), f, n, 1 if i == 0 else 2))
    return {"stem": stem, "stacks": out}


def conv_names(spec):
    """Every conv/bn pair in forward order: (name, kh, cin, cout, stride)."""
    names = [("conv1", 7, 3, spec["stem"], 2)]
    cin = spec["stem"]
    for sname, f, n, s1 in spec["stacks"]:
        for b in range(1, n + 1):
            p = "%s_block%d" % (sname, b)
            stride = s1 if b == 1 else 1
            if b == 1:
                names.append((p + "_0", 1, cin, 4 * f, stride))
            names.append((p + "_1", 1, cin, f, stride))
            names.append((p + "_2", 3, f, f, 1))
            names.append((p + "_3", 1, f, 4 * f, 1))
            cin = 4 * f
    return names


def _t(a, dtype):
    return torch.as_tensor(np.ascontiguousarray(a)).to(dtype)


def _w(W_hwio, dtype):
    return _t(W_hwio, dtype).permute(3, 2, 0, 1).contiguous()


def _bn(x, w, name, dtype):
    g, b, mu, var = (_t(w[name + "_bn_" + k], dtype).view(1, -1, 1, 1) for k in ("gamma", "beta", "mean", "var"))
    return g * (x - mu) / torch.sqrt(var + BN_EPS) + b


class _Tape(object):
    """Forward pass that records, per layer, what the reverse rules need."""

    def __init__(self, w, spec, dtype):
        self.w, self.spec, self.dtype = w, spec, dtype
        self.ops = []      # (kind, payload) in forward order

    def conv(self, x, name, stride, pad):
        W, b = _w(self.w[name + "_conv_W"], self.dtype), _t(self.w[name + "_conv_b"], self.dtype)
        y = F.conv2d(x, W, b, stride=stride, padding=pad)
        self.ops.append(("conv", (x, W, b, stride, pad)))
        return y

    def bn(self, x, name):
        y = _bn(x, self.w, name, self.dtype)
        self.ops.append(("bn", (x, y, name)))
        return y

    def forward(self, X_nhwc):
        x = _t(X_nhwc, self.dtype).permute(0, 3, 1, 2).contiguous()
        xp = F.pad(x, (3, 3, 3, 3))
        self.ops.append(("pad", 3))
        y = self.bn(self.conv(xp, "conv1", 2, 0), "conv1")
        a = F.relu(y)
        self.ops.append(("relu", None))
        ap = F.pad(a, (1, 1, 1, 1))
        self.ops.append(("pad", 1))
        p = F.max_pool2d(ap, 3, 2)
        self.ops.append(("pool", ap))
        t = p
        for sname, f, n, s1 in self.spec["stacks"]:
            for b in range(1, n + 1):
                nm = "%s_block%d" % (sname, b)
                stride = s1 if b == 1 else 1
                self.ops.append(("fork", None))                 # t feeds main path and shortcut
                if b == 1:
                    self.ops.append(("branch", "shortcut"))
                    sc = self.bn(self.conv(t, nm + "_0", stride, 0), nm + "_0")
                    self.ops.append(("endbranch", "shortcut"))
                else:
                    sc = t
                self.ops.append(("branch", "main"))
                y1 = F.relu(self.bn(self.conv(t, nm + "_1", stride, 0), nm + "_1"))
                self.ops.append(("relu", None))
                y2 = F.relu(self.bn(self.conv(y1, nm + "_2", 1, 1), nm + "_2"))
                self.ops.append(("relu", None))
                y3 = self.bn(self.conv(y2, nm + "_3", 1, 0), nm + "_3")
                self.ops.append(("endbranch", "main"))
                self.ops.append(("add", (sc, y3, b == 1)))
                t = F.relu(sc + y3)
                self.ops.append(("relu", None))
        return t


def forward(w, spec, X_nhwc, dtype=torch.float64):
    tape = _Tape(w, spec, dtype)
    return tape.forward(X_nhwc).permute(0, 2, 3, 1).contiguous().numpy()


def _alpha1beta0(x, W, b, stride, pad, R, dtype):
    """RR:274-322 for a general Conv2D (stride / padding / kernel size)."""
    wp, wn = W * (W >= 0).to(dtype), W * (W < 0).to(dtype)
    bp, bn = b * (b >= 0).to(dtype), b * (b < 0).to(dtype)
    x1 = (x * (x >= 0).to(dtype)).detach().requires_grad_(True)
    x2 = (x * (x < 0).to(dtype)).detach().requires_grad_(True)
    Z1 = F.conv2d(x1, wp, bp, stride=stride, padding=pad)
    Z2 = F.conv2d(x2, wn, bn, stride=stride, padding=pad)
    S = safe_divide(R, (Z1 + Z2).detach())
    g1, = torch.autograd.grad(Z1, x1, grad_outputs=S)
    g2, = torch.autograd.grad(Z2, x2, grad_outputs=S)
    return (x1 * g1 + x2 * g2).detach()


def _bn_reverse(x, y, name, w, R, dtype):
    """RA:197-257."""
    beta = _t(w[name + "_bn_beta"], dtype).view(1, -1, 1, 1)
    mu = _t(w[name + "_bn_mean"], dtype).view(1, -1, 1, 1)
    num = x * (y - beta) * R
    den = (x - mu) * y
    den = den + ((den >= 0).to(dtype) * 2 - 1) * SAFE_EPS
    return safe_divide(num, den)


def analyze(w, spec, X_nhwc, R_nhwc, dtype=torch.float64):
    """(N,H,W,3), (N,h,w,C) -> (N,H,W,3): the literal reverse walk."""
    tape = _Tape(w, spec, dtype)
    tape.forward(X_nhwc)
    R = _t(R_nhwc, dtype).permute(0, 3, 1, 2).contiguous()
    # reverse interpretation of the tape with an explicit stack for the residual forks
    ops = tape.ops
    i = len(ops) - 1
    pending = []          # stack of dicts for open Add layers: {"sc": R, "main": R, ...}
    while i >= 0:
        kind, p = ops[i]
        if kind == "relu":
            pass                                                     # RA:462-469
        elif kind == "add":
            sc, y3, _ = p
            S = safe_divide(R, sc + y3)                               # RA:260-286
            pending.append({"R_sc": sc * S, "R_main": y3 * S, "acc": None})
            R = None
        elif kind == "endbranch":
            R = pending[-1]["R_main"] if p == "main" else pending[-1]["R_sc"]
        elif kind == "branch":
            d = pending[-1]
            d["acc"] = R if d["acc"] is None else d["acc"] + R        # relevance reaching the fork via this branch
            if p == "shortcut":
                d["had_sc_branch"] = True
            R = None
        elif kind == "fork":
            d = pending.pop()
            if not d.get("had_sc_branch"):
                d["acc"] = d["acc"] + d["R_sc"]                       # identity shortcut: R_sc lands on t directly
            R = d["acc"]                                              # KG:799-803
        elif kind == "conv":
            x, W, b, stride, pad = p
            R = _alpha1beta0(x, W, b, stride, pad, R, dtype)
        elif kind == "bn":
            x, y, name = p
            R = _bn_reverse(x, y, name, tape.w, R, dtype)
        elif kind == "pool":
            ap = p.detach().requires_grad_(True)
            g, = torch.autograd.grad(F.max_pool2d(ap, 3, 2), ap, grad_outputs=R)     # RA:470-480
            R = g.detach()
        elif kind == "pad":
            R = R[:, :, p:-p, p:-p]                                   # gradient of ZeroPadding2D = crop
        i -= 1
    return R.permute(0, 2, 3, 1).contiguous().numpy()


def analyze_cached(w, spec, X_nhwc, R_nhwc, dtype=torch.float64):
    """The restructured algorithm of the HIP ResNet path, float64 on CPU (equals `analyze` to round-off,
    tests/test_oracle_resnet.py).  Per image and conv l (input x_l >= 0 except at the stem):
        c_l = conv(x_l, w_l) + b_l ;  Z_l = conv(x_l, w_l+) + b_l ;  y_l = BN(c_l)
        Q_l = c_l (y_l - beta_l) / stab((c_l - mu_l) y_l) / safe(Z_l)      [BN reverse o alpha1beta0 denominator]
    per block with input t, shortcut sc, main output y3, o = relu(sc + y3):
        fA = y3 / safe(sc + y3), fS = sc / safe(sc + y3)
    per token:  S3 = R_o fA Q3 -> C3 = convT(S3, w3+) -> S2 = C3 (a2 Q2) -> C2 -> S1 = C2 (a1 Q1) -> C1
                R_t = t C1 + ( identity: R_o fS | projection: t convT(R_o fS Q0, w0+) )."""
    x = _t(X_nhwc, dtype).permute(0, 3, 1, 2).contiguous()

    def stab(d):
        return d + ((d >= 0).to(dtype) * 2 - 1) * SAFE_EPS

    def unit(t_in, name, stride, pad, both_signs=False):
        W, b = _w(w[name + "_conv_W"], dtype), _t(w[name + "_conv_b"], dtype)
        wp, wn = W * (W >= 0), W * (W < 0)
        c = F.conv2d(t_in, W, b, stride=stride, padding=pad)
        if both_signs:
            Z = F.conv2d(t_in * (t_in >= 0), wp, None, stride=stride, padding=pad) + \
                F.conv2d(t_in * (t_in < 0), wn, None, stride=stride, padding=pad) + b.view(1, -1, 1, 1)
        else:
            Z = F.conv2d(t_in, wp, b, stride=stride, padding=pad)
        y = _bn(c, w, name, dtype)
        beta = _t(w[name + "_bn_beta"], dtype).view(1, -1, 1, 1)
        mu = _t(w[name + "_bn_mean"], dtype).view(1, -1, 1, 1)
        Q = safe_divide(c * (y - beta), stab((c - mu) * y)) / (Z + (Z == 0) * SAFE_EPS)
        return y, Q, wp, wn

    xp = F.pad(x, (3, 3, 3, 3))
    y0, Q0, wp0, wn0 = unit(xp, "conv1", 2, 0, both_signs=True)
    a0 = F.relu(y0)
    ap = F.pad(a0, (1, 1, 1, 1))
    t = F.max_pool2d(ap, 3, 2)
    blocks = []
    for sname, f, n, s1 in spec["stacks"]:
        for b in range(1, n + 1):
            nm = "%s_block%d" % (sname, b)
            stride = s1 if b == 1 else 1
            d = {"t": t, "stride": stride, "proj": b == 1}
            if b == 1:
                sc, d["Q0"], d["w0p"], _ = unit(t, nm + "_0", stride, 0)
            else:
                sc = t
            y1, d["Q1"], d["w1p"], _ = unit(t, nm + "_1", stride, 0)
            a1 = F.relu(y1)
            y2, d["Q2"], d["w2p"], _ = unit(a1, nm + "_2", 1, 1)
            a2 = F.relu(y2)
            y3, d["Q3"], d["w3p"], _ = unit(a2, nm + "_3", 1, 0)
            den = sc + y3
            den = den + (den == 0) * SAFE_EPS
            d["fA"], d["fS"], d["a1"], d["a2"] = y3 / den, sc / den, a1, a2
            blocks.append(d)
            t = F.relu(sc + y3)
    R = _t(R_nhwc, dtype).permute(0, 3, 1, 2).contiguous()
    for d in reversed(blocks):
        s = d["stride"]
        S3 = R * d["fA"] * d["Q3"]
        S2 = F.conv_transpose2d(S3, d["w3p"]) * (d["a2"] * d["Q2"])
        S1 = F.conv_transpose2d(S2, d["w2p"], padding=1) * (d["a1"] * d["Q1"])
        op = s - 1                                       # output_padding restores the fine resolution of a stride-2 conv
        Rt = d["t"] * F.conv_transpose2d(S1, d["w1p"], stride=s, output_padding=op)
        if d["proj"]:
            Rt = Rt + d["t"] * F.conv_transpose2d(R * d["fS"] * d["Q0"], d["w0p"], stride=s, output_padding=op)
        else:
            Rt = Rt + R * d["fS"]
        R = Rt
    # pool routing (overlapping 3x3/2 windows on the 1-padded map), then the stem
    apr = ap.detach().requires_grad_(True)
    g, = torch.autograd.grad(F.max_pool2d(apr, 3, 2), apr, grad_outputs=R)
    Ra0 = g[:, :, 1:-1, 1:-1]
    S0 = Ra0 * Q0
    xpp, xpn = xp * (xp >= 0), xp * (xp < 0)
    Rx = xpp * F.conv_transpose2d(S0, wp0, stride=2, output_padding=1) + xpn * F.conv_transpose2d(S0, wn0, stride=2, output_padding=1)
    return Rx[:, :, 3:-3, 3:-3].permute(0, 2, 3, 1).contiguous().numpy()
