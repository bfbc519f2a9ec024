"""CPU: where does the reference's float32 ResNet walk leave the float64 one on the COCO photograph?"""
import os, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
import torch.nn.functional as F
from oracle import resnet_lrp_ref as RN
from oracle.resnet_lrp_ref import _Tape, _t, _alpha1beta0, _bn_reverse, safe_divide
from oracle.decoder_ref import GridTDOracle
from lrp_imagecaptioning_amd.synthetic import captions, gridtd_weights, resnet_weights
torch.set_num_threads(8)
rgb = np.load("/root/repo/tests/golden/real_images.npz")["rgb_u8"][2:3]
rs = np.random.RandomState(4)
V, T = 1000, 6
w = resnet_weights(rs)
w.update(gridtd_weights(rs, 49, 2048, 512, 512, V))
spec = RN.resnet_spec()
cap = captions(rs, 1, T, V)[0]
Xh = np.ascontiguousarray(rgb.astype(np.float32)[..., ::-1] - np.array([103.939, 116.779, 123.68], dtype=np.float32))
feat = RN.forward(w, spec, Xh)
o = GridTDOracle(w, 49, 2048, 512, 512)
o.forward(feat.astype(np.float32), cap)
Rf = o.explain(3)[0].reshape(1, 7, 7, 2048)

def walk(dtype):
    tape = _Tape(w, spec, dtype)
    tape.forward(Xh)
    R = _t(Rf, dtype).permute(0, 3, 1, 2).contiguous()
    ops = tape.ops
    i = len(ops) - 1
    pending = []
    trace = []
    while i >= 0:
        kind, p = ops[i]
        if kind == "add":
            sc, y3, _ = p
            S = safe_divide(R, sc + y3)
            pending.append({"R_sc": sc * S, "R_main": y3 * S, "acc": None})
            R = None
        elif kind == "endbranch":
            R = pending[-1]["R_main"] if p == "main" else pending[-1]["R_sc"]
        elif kind == "branch":
            d = pending[-1]
            d["acc"] = R if d["acc"] is None else d["acc"] + R
            if p == "shortcut":
                d["had_sc_branch"] = True
            R = None
        elif kind == "fork":
            d = pending.pop()
            if not d.get("had_sc_branch"):
                d["acc"] = d["acc"] + d["R_sc"]
            R = d["acc"]
        elif kind == "conv":
            x, W, b, stride, pad = p
            R = _alpha1beta0(x, W, b, stride, pad, R, dtype)
        elif kind == "bn":
            x, y, name = p
            R = _bn_reverse(x, y, name, tape.w, R, dtype)
            trace.append((i, "bn " + name, R.double().clone(), (x.double(), y.double())))
        elif kind == "pool":
            ap = p.detach().requires_grad_(True)
            g, = torch.autograd.grad(F.max_pool2d(ap, 3, 2), ap, grad_outputs=R)
            R = g.detach()
        elif kind == "pad":
            R = R[:, :, p:-p, p:-p]
        if kind == "fork":
            trace.append((i, "fork", R.double().clone(), None))
        i -= 1
    return R, trace

R64, t64 = walk(torch.float64)
R32, t32 = walk(torch.float32)
print("final rel L1 %.3e" % float((R32.double() - R64).abs().sum() / R64.abs().sum()))
prev = 0.0
for (i, nm, a, xy), (_, _, b, _) in zip(t64, t32):
    e = float((a - b).abs().sum() / a.abs().sum())
    flag = " <<<" if e > 3 * max(prev, 1e-7) and e > 5e-6 else ""
    if flag or nm == "fork" and False:
        print("%4d %-28s rel L1 %.3e (before %.3e)%s" % (i, nm, e, prev, flag))
        if xy is not None:
            x, y = xy
            beta = torch.as_tensor(w[nm[3:] + "_bn_beta"], dtype=torch.float64).view(1, -1, 1, 1)
            mu = torch.as_tensor(w[nm[3:] + "_bn_mean"], dtype=torch.float64).view(1, -1, 1, 1)
            den = (x - mu) * y
            d = (a - b).abs()
            k = int(d.flatten().argmax())
            idx = np.unravel_index(k, d.shape)
            print("      largest difference at", idx, "R64 %.4e R32 %.4e  c-mu %.3e  y %.3e  den %.3e; share of the layer's difference %.2f; |den| < 1e-5 at %d of %d units"
                  % (float(a[idx]), float(b[idx]), float((x - mu)[idx]), float(y[idx]), float(den[idx]), float(d[idx] / d.sum()), int((den.abs() < 1e-5).sum()), den.numel()))
    prev = e
