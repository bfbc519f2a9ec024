import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from conftest import rel_l1
from lrp_imagecaptioning_amd.engine import LRPEngine, preprocess_images, switches
from lrp_imagecaptioning_amd.synthetic import RESNET101_STACKS, captions, gridtd_weights, images, resnet_weights
from oracle import resnet_lrp_ref as RN
from oracle.decoder_ref import GridTDOracle
rgb_all = np.load("tests/golden/real_images.npz")["rgb_u8"]
rs = np.random.RandomState(4)
V, T = 1000, 6
w = resnet_weights(rs)
w.update(gridtd_weights(rs, 49, 2048, 512, 512, V))
spec = RN.resnet_spec()
cap = captions(rs, 1, T, V)[0]
kw = dict(decoder="gridtd", img_hw=(224, 224), L=49, D=2048, H=512, E=512, V=V, max_caption_len=T + 1,
          resnet={"stem": 64, "stacks": RESNET101_STACKS})
tt = [1, 3, T]
noise = images(np.random.RandomState(9), 1)
for name, Xh in [("photo2", rgb_all[2:3].astype(np.float32)[..., ::-1] - np.array([103.939, 116.779, 123.68], dtype=np.float32)),
                 ("photo0", rgb_all[0:1].astype(np.float32)[..., ::-1] - np.array([103.939, 116.779, 123.68], dtype=np.float32)),
                 ("noise", noise)]:
    Xh = np.ascontiguousarray(Xh)
    feat_ref = RN.forward(w, spec, Xh)
    o = GridTDOracle(w, 49, 2048, 512, 512)
    o.forward(feat_ref.astype(np.float32), cap)
    Rs = [o.explain(t)[0].reshape(1, 7, 7, 2048) for t in tt]
    refs = [RN.analyze(w, spec, Xh, R)[0] for R in Rs]
    refs32 = [RN.analyze(w, spec, Xh, R, dtype=torch.float32)[0] for R in Rs]
    print(name, "float32 literal graph vs float64:", ["%.2e" % rel_l1(a, b) for a, b in zip(refs32, refs)], flush=True)
    for prec, emit in (("bf16x3", 1), ("bf16x3", 0), ("fp32", 1)):
        with switches(LRP_FWD_EMIT=emit):
            e = LRPEngine(max_images=1, max_tokens=T, **kw)
            e.set_precision(prec)
            e.set_weights(w)
            Xd = torch.as_tensor(Xh).cuda()
            e.encode_images(Xd)
            feat = e.get_features().cpu().numpy().reshape(feat_ref.shape)
            Rcat = np.concatenate(Rs).astype(np.float32)
            out_given = e.cnn_explain([0, 0, 0], Rcat).cpu().numpy()      # CNN half alone, on the oracle's R
            e.decoder_forward([cap])
            out = e.explain_tokens([0] * 3, tt)[0].cpu().numpy()
        print("  %s emit=%d features %.2e  cnn-only %s  end-to-end %s" % (prec, emit, rel_l1(feat, feat_ref),
              ["%.2e" % rel_l1(out_given[i], refs[i]) for i in range(3)], ["%.2e" % rel_l1(out[i], refs[i]) for i in range(3)]), flush=True)
