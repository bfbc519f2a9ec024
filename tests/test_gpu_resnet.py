"""-m gpu: ResNet (v1 bottleneck) encoder LRP through the C ABI against the float64 literal oracle
(oracle/resnet_lrp_ref.py) — BASELINE config 4's CNN half (rows c3, c5, c6 of SURVEY §8a)."""
import numpy as np
import pytest

from conftest import rel_l1
from gpu_util import report
from lrp_imagecaptioning_amd.synthetic import RESNET101_STACKS, gridtd_weights, resnet_weights
from oracle import resnet_lrp_ref as RN
from oracle.decoder_ref import GridTDOracle

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _engine(stacks, stem, hw, B, ntok, w, decoder="gridtd", H=32, V=50):
    from lrp_imagecaptioning_amd.engine import LRPEngine
    side = hw // 4 // (2 ** (len(stacks) - 1))
    D = 4 * stacks[-1][0]
    eng = LRPEngine(decoder=decoder, img_hw=(hw, hw), L=side * side, D=D, H=H, E=H, V=V, max_images=B, max_tokens=ntok,
                    max_caption_len=6, resnet={"stem": stem, "stacks": stacks})
    eng.set_weights(w)
    return eng, side, D


@pytest.mark.parametrize("prec", ["bf16x3", "fp32"])
@pytest.mark.parametrize("name,stacks,stem,hw,B", [("tiny", ((4, 2), (8, 2)), 8, 32, 2),      # widths % 8 != 0: fp32 either way
                                                   ("mid", ((8, 2), (16, 3), (32, 2)), 16, 64, 2),
                                                   # 64-channel stem on a 96 x 96 image: the fused stem reverse (rn_stem_reverse_kernel:
                                                   # 4 x 4 patches of the 48 x 48 stem map, the last ones ragged), pair-emitting forward
                                                   ("stem64", ((32, 2), (64, 2)), 64, 96, 2)])
def test_small_resnets_match_oracle(name, stacks, stem, hw, B, prec):
    rs = np.random.RandomState(3)
    w = resnet_weights(rs, stacks, stem=stem, bias_std=0.2)
    spec = RN.resnet_spec(stacks, stem=stem)
    X = rs.uniform(-120, 130, size=(B, hw, hw, 3)).astype(np.float32)
    eng, side, D = _engine(stacks, stem, hw, B, 2 * B, w)
    eng.set_precision(prec)
    eng.encode_images(X)
    feat = eng.get_features().cpu().numpy().reshape(B, side, side, D)
    feat_ref = RN.forward(w, spec, X)
    assert rel_l1(feat, feat_ref) < 1e-5
    idx = list(range(B)) + list(range(B))[::-1]
    R = (rs.standard_normal((2 * B,) + feat_ref.shape[1:]) * feat_ref[idx]).astype(np.float32)
    out = eng.cnn_explain(idx, R).cpu().numpy()
    ref = RN.analyze(w, spec, X[idx], R)
    errs = [rel_l1(out[i], ref[i]) for i in range(2 * B)]
    report("resnet_" + name, prec=prec, feat_rel_l1=rel_l1(feat, feat_ref), max_rel_l1=max(errs))
    assert np.isfinite(out).all()
    assert max(errs) < TOL, errs


@pytest.mark.parametrize("prec", ["bf16x3", "fp32"])
def test_resnet101_full_size_matches_oracle(prec):
    """ResNet-101, 224x224 -> (7,7,2048) (config.py:41-45), one image, two relevance maps; the reverse walk's conv
    chains in the default split-bf16 mode and in exact fp32."""
    rs = np.random.RandomState(0)
    w = resnet_weights(rs)
    spec = RN.resnet_spec()
    X = rs.uniform(-120, 130, size=(1, 224, 224, 3)).astype(np.float32)
    eng, side, D = _engine(RESNET101_STACKS, 64, 224, 1, 2, w)
    assert (side, D) == (7, 2048)
    eng.set_precision(prec)
    eng.encode_images(X)
    feat = eng.get_features().cpu().numpy().reshape(1, 7, 7, 2048)
    feat_ref = RN.forward(w, spec, X)
    e_feat = rel_l1(feat, feat_ref)
    R = (rs.standard_normal((2, 7, 7, 2048)) * feat_ref).astype(np.float32)
    out = eng.cnn_explain([0, 0], R).cpu().numpy()
    ref = RN.analyze(w, spec, np.repeat(X, 2, axis=0), R)
    errs = [rel_l1(out[i], ref[i]) for i in range(2)]
    report("resnet101", prec=prec, feat_rel_l1=e_feat, max_rel_l1=max(errs))
    assert e_feat < 1e-5
    assert max(errs) < TOL, errs


def test_pair_emitting_forward_matches_oracle_is_batch_invariant_and_has_a_fallback():
    """Round 4's ResNet forward (resnet_encoder.h encode_emit: every unit's epilogue writes the next conv's fp16 pairs, the block-end
    kernel and the stem's pool the block inputs', scales per image from bounds on measured maxima) on a network whose every
    unit qualifies (widths % 32 == 0: two stacks, a stride-1 and a stride-2 projection block, identity blocks, 3 images):
    features and heat-maps vs the float64 oracle; an image alone == the same image inside the batch, bit for bit (what the
    per-call scales of round 3 could not give); LRP_FWD_EMIT=0 (split passes between the convs) agrees to the forward's rounding;
    the walk's projection-block path (S3 from the previous epilogue, join + next head in one pass) is the same arithmetic."""
    import torch
    from lrp_imagecaptioning_amd.engine import switches
    stacks, stem, hw, B = ((32, 3), (64, 2)), 32, 64, 3
    rs = np.random.RandomState(12)
    w = resnet_weights(rs, stacks, stem=stem, bias_std=0.2)
    spec = RN.resnet_spec(stacks, stem=stem)
    X = rs.uniform(-120, 130, size=(B, hw, hw, 3)).astype(np.float32)
    X[2] *= 0.01                                           # an image 100x dimmer than its batch mates
    eng, side, D = _engine(stacks, stem, hw, B, 2 * B, w)
    one, _, _ = _engine(stacks, stem, hw, 1, 2, w)
    feat_ref = RN.forward(w, spec, X)
    idx = [0, 1, 2, 2, 1, 0]
    R = (rs.standard_normal((2 * B,) + feat_ref.shape[1:]) * feat_ref[idx]).astype(np.float32)
    ref = RN.analyze(w, spec, X[idx], R)

    def run(e, Xs, ix, Rs):
        e.encode_images(Xs)
        return e.get_features().clone(), e.cnn_explain(ix, Rs).clone()
    feat, out = run(eng, X, idx, R)
    e_feat = rel_l1(feat.cpu().numpy().reshape(feat_ref.shape), feat_ref)
    errs = [rel_l1(out[i].cpu().numpy(), ref[i]) for i in range(2 * B)]
    for n in range(B):                                     # batch invariance
        rows = [i for i in range(2 * B) if idx[i] == n]
        f1, o1 = run(one, X[n:n + 1], [0, 0], R[rows])
        assert torch.equal(f1[0], feat[n]), n
        assert torch.equal(o1, out[rows]), n
    with switches(LRP_FWD_EMIT=0):
        feat0, out0 = run(eng, X, idx, R)
    e_fb = float((feat0.double() - feat.double()).abs().sum() / feat.double().abs().sum())
    e_fb_hm = max(rel_l1(out0[i].cpu().numpy(), out[i].cpu().numpy()) for i in range(2 * B))
    report("resnet_emit", feat_rel_l1=e_feat, max_rel_l1=max(errs), fallback_feat=e_fb, fallback_heatmaps=e_fb_hm)
    assert e_feat < 1e-5
    assert max(errs) < TOL, errs
    assert e_fb < 1e-5 and e_fb_hm < 2e-5


def _stem_pool_windows(w, Xh, dtype):
    """candidates of every window of the stem's overlapping 3x3/2 max-pool (zero padding included): (1, 64, 9, 56*56)"""
    import torch
    import torch.nn.functional as F
    x = torch.as_tensor(Xh.copy(), dtype=dtype).permute(0, 3, 1, 2)
    c = F.conv2d(F.pad(x, (3, 3, 3, 3)), torch.as_tensor(w["conv1_conv_W"], dtype=dtype).permute(3, 2, 0, 1).contiguous(),
                 torch.as_tensor(w["conv1_conv_b"], dtype=dtype), stride=2)
    g, b_, m, v = (torch.as_tensor(w["conv1_bn_" + k], dtype=dtype).view(1, -1, 1, 1) for k in ("gamma", "beta", "mean", "var"))
    a0 = torch.relu(g * (c - m) / torch.sqrt(v + 1.001e-5) + b_)
    return F.unfold(F.pad(a0, (1, 1, 1, 1)), 3, stride=2).view(1, 64, 9, -1)


@pytest.mark.parametrize("pic", [0, 2], ids=["flickr30k_1009434119", "coco_000000005586"])
def test_a_photograph_through_resnet101_and_gridtd(pic):
    """BASELINE config 4's path on NATURAL images (tests/golden/real_images.npz: photographs the reference ships): device
    preprocessing -> ResNet-101 -> grid-TD -> three per-word heat-maps vs the oracles; and the same picture inside a batch of
    noise images, bit for bit.  What a photograph adds over noise images here:
      * tens of thousands of windows of the stem's overlapping 3x3/2 max-pool are all zero (a tie of every candidate, the zero
        padding included; counted in the float64 forward and asserted);
      * NEAR-ties: the COCO picture has a flat region, and there two windows have their two best candidates within 1e-6 of
        each other — and in ONE of them the reference's own float32 forward picks the other candidate than float64 does (counted
        below from both forwards).  `MaxPoolGrad` then routes that window's relevance to another pixel: the float32 literal
        graph (what the reference's TensorFlow computes) is 2.7e-5 / 1.0e-4 / 5.0e-5 from the float64 one on the three words,
        44 % of it at that one unit, and every arithmetic of the engine (the exact fp32 mode included) lands 0.9 - 1.5e-4 from
        float64 — a discrete decision of the reference's algorithm on a quantity float32 cannot order, not rounding that more
        bits in the walk would remove (profiles/r04_resnet_photograph.txt).  Bound for a picture with such windows:
        max(1e-4, 3 x the float32 literal graph's largest distance), as in test_gpu_stress_parity.py; without: 1e-4."""
    import os
    import torch
    from conftest import GOLDEN
    from lrp_imagecaptioning_amd.engine import LRPEngine, preprocess_images
    from lrp_imagecaptioning_amd.synthetic import captions, images
    rgb = np.load(os.path.join(GOLDEN, "real_images.npz"))["rgb_u8"][pic:pic + 1]
    rs = np.random.RandomState(4)
    V, T = 1000, 6
    w = resnet_weights(rs)
    w.update(gridtd_weights(rs, 49, 2048, 512, 512, V))
    spec = RN.resnet_spec()
    cap = captions(rs, 1, T, V)[0]
    Xh = np.ascontiguousarray(rgb.astype(np.float32)[..., ::-1] - np.array([103.939, 116.779, 123.68], dtype=np.float32))   # preprocessors.py:38-53
    w64, w32 = _stem_pool_windows(w, Xh, torch.float64), _stem_pool_windows(w, Xh, torch.float32)
    top2 = w64.topk(2, dim=2)[0]
    dead = int((top2[:, :, 0] == 0).sum())
    near = int(((top2[:, :, 0] > 0) & ((top2[:, :, 0] - top2[:, :, 1]) < 1e-6 * top2[:, :, 0])).sum())
    flips32 = int(((w64.argmax(2) != w32.argmax(2)) & (top2[:, :, 0] > 0)).sum())
    kw = dict(decoder="gridtd", img_hw=(224, 224), L=49, D=2048, H=512, E=512, V=V, max_caption_len=T + 1,
              resnet={"stem": 64, "stacks": RESNET101_STACKS})
    one = LRPEngine(max_images=1, max_tokens=T, **kw)
    one.set_weights(w)
    Xd = preprocess_images(torch.as_tensor(rgb).cuda())
    assert torch.equal(Xd.cpu(), torch.as_tensor(Xh))
    one.encode_images(Xd)
    one.decoder_forward([cap])
    tt = [1, 3, T]
    out = one.explain_tokens([0] * 3, tt)[0].clone()
    o = GridTDOracle(w, 49, 2048, 512, 512)
    o.forward(RN.forward(w, spec, Xh).astype(np.float32), cap)
    Rs = [o.explain(t)[0].reshape(1, 7, 7, 2048) for t in tt]
    refs = [RN.analyze(w, spec, Xh, R)[0] for R in Rs]
    errs = [rel_l1(out[i].cpu().numpy(), refs[i]) for i in range(3)]
    noise32 = 0.0
    if near:                                               # the reference's own float32 arithmetic on this picture
        noise32 = max(rel_l1(RN.analyze(w, spec, Xh, R, dtype=torch.float32)[0], ref) for R, ref in zip(Rs, refs))
    B = 4
    big = LRPEngine(max_images=B, max_tokens=3, **kw)
    big.set_weights(w)
    Xb = torch.as_tensor(images(rs, B)).cuda()
    Xb[2] = Xd[0]
    caps = captions(rs, B, T, V)
    caps[2] = cap
    big.encode_images(Xb)
    big.decoder_forward(caps)
    same = bool(torch.equal(big.explain_tokens([2] * 3, tt)[0], out))
    report("resnet101_photograph_%d" % pic, stem_pool_all_zero_windows=dead, stem_pool_near_ties=near,
           float32_forward_picks_another_candidate=flips32, float32_literal_graph_rel_l1=noise32, rel_l1=errs, in_batch_bit_identical=same)
    assert dead >= 10000, dead
    assert max(errs) < max(TOL, 3 * noise32), (errs, noise32)
    assert same


def test_resnet101_walk_is_linear_zero_preserving_and_reproducible():
    """Size-independent properties of the ResNet walk at full size (no oracle needed): analyze(3 R1 - 2 R2) = 3 analyze(R1) -
    2 analyze(R2) (the walk is a linear map once the forward's gates are fixed: what is left is the rounding of the split-bf16
    operands), zero relevance in -> exactly zero out, powers of two scale exactly, and a second call returns the same bits
    (no atomics on the data path: the fused stem reverse and the fixed-order joins included)."""
    import torch
    rs = np.random.RandomState(6)
    w = resnet_weights(rs)
    X = rs.uniform(-120, 130, size=(2, 224, 224, 3)).astype(np.float32)
    eng, side, D = _engine(RESNET101_STACKS, 64, 224, 2, 6, w)
    eng.encode_images(X)
    feat = eng.get_features().cpu().numpy().reshape(2, side, side, D)
    Ra = (rs.standard_normal(feat[1].shape) * feat[1]).astype(np.float32)
    Rb = (rs.standard_normal(feat[1].shape) * feat[1]).astype(np.float32)
    R = np.stack([Ra, Rb, 3 * Ra - 2 * Rb, np.zeros_like(Ra), 4.0 * Ra, (rs.standard_normal(feat[0].shape) * feat[0]).astype(np.float32)])
    idx = [1, 1, 1, 1, 1, 0]
    out = eng.cnn_explain(idx, R).clone()
    again = eng.cnn_explain(idx, R)
    assert torch.equal(out, again)
    o = out.cpu().numpy()
    lin = rel_l1(o[2], 3 * o[0] - 2 * o[1])
    from lrp_imagecaptioning_amd.engine import switches
    with switches(LRP_IMG_FUSED=0):                       # the stem's reverse as 1-tap GEMM + gather kernel: the same sums, another order
        unfused = eng.cnn_explain(idx, R).cpu().numpy()
    stem_ab = max(rel_l1(o[i], unfused[i]) for i in (0, 1, 5))
    report("resnet101_linearity", rel_l1=lin, fused_stem_vs_two_kernels=stem_ab)
    assert stem_ab < 1e-6
    assert lin < 2e-5
    assert (o[3] == 0).all()
    assert np.array_equal(o[4], 4.0 * o[0])
    assert np.isfinite(o).all()


def test_config4_gridtd_plus_resnet_end_to_end():
    """grid-TD decoder on a ResNet encoder (BASELINE config 4 at reduced size): decoder LRP -> CNN LRP fused call."""
    stacks, stem, hw, H, V = ((4, 2), (8, 2)), 8, 32, 32, 50
    rs = np.random.RandomState(9)
    w = resnet_weights(rs, stacks, stem=stem, bias_std=0.2)
    L, D = 16, 32
    w.update(gridtd_weights(rs, L, D, H, H, V))
    X = rs.uniform(-120, 130, size=(1, hw, hw, 3)).astype(np.float32)
    cap = [7, 19, 33, 1]
    eng, side, _ = _engine(stacks, stem, hw, 1, 4, w, decoder="gridtd", H=H, V=V)
    eng.encode_images(X)
    eng.decoder_forward([cap])
    out, _, _, _ = eng.explain_tokens([0, 0, 0], [1, 2, 3])
    out = out.cpu().numpy()
    spec = RN.resnet_spec(stacks, stem=stem)
    o = GridTDOracle(w, L, D, H, H)
    o.forward(RN.forward(w, spec, X).astype(np.float32), cap)
    worst = max(rel_l1(out[t - 1], RN.analyze(w, spec, X, o.explain(t)[0])[0]) for t in (1, 2, 3))
    report("config4_small", max_rel_l1=worst)
    assert worst < TOL


def test_reference_surface_with_resnet():
    """ExplainImgCaptioningGridTDModel on a ResNet spec, and LRPSequentialPresetA.analyze([X, R]) with the
    (h, w, 4f) head the reference hard-codes as (7,7,2048) (base.py:370-373)."""
    from lrp_imagecaptioning_amd.analyzer import ImageModelSpec, LRPSequentialPresetA
    from lrp_imagecaptioning_amd.explainers import CaptionModelSpec, ExplainImgCaptioningGridTDModel
    stacks, stem, hw, H, V, L, D = ((4, 2), (8, 2)), 8, 32, 32, 50, 16, 32
    rs = np.random.RandomState(2)
    w = resnet_weights(rs, stacks, stem=stem, bias_std=0.2)
    w.update(gridtd_weights(rs, L, D, H, H, V))
    rn = {"stem": stem, "stacks": stacks}
    X = rs.uniform(-120, 130, size=(1, hw, hw, 3)).astype(np.float32)
    spec = RN.resnet_spec(stacks, stem=stem)
    an = LRPSequentialPresetA(ImageModelSpec(w, img_hw=(hw, hw), resnet=rn), epsilon=0.01, neuron_selection_mode="replace")
    feat = RN.forward(w, spec, X)
    R = (rs.standard_normal(feat.shape) * feat).astype(np.float32)
    assert rel_l1(an.analyze([X, R]), RN.analyze(w, spec, X, R)) < TOL
    ex = ExplainImgCaptioningGridTDModel(
        CaptionModelSpec(w, img_encoder="resnet101", hidden_dim=H, embedding_dim=H, L=L, D=D, vocab_size=V, img_hw=(hw, hw),
                         resnet=rn), None, None, max_caption_length=5)
    cap = [9, 21, 1]
    ex._forward_beam_search((None, X), cap)
    rel, att = ex._explain_sentence()
    o = GridTDOracle(w, L, D, H, H)
    o.forward(feat.astype(np.float32), cap)
    for i, Rf in enumerate(rel):
        assert rel_l1(ex._explain_CNN(X, Rf), RN.analyze(w, spec, X, o.explain(i + 1)[0])) < TOL


def test_resnet_handle_takes_weights_from_device():
    """lrp_set_weight_dev on a ResNet handle: the decoder's weights are packed on the device, the encoder units (conv + BN
    folding is a host packer) are staged once through the host — the handle must end up in the host-set state exactly."""
    import torch
    stacks, stem, hw, H, V = ((4, 2), (8, 2)), 8, 32, 32, 50
    rs = np.random.RandomState(19)
    w = resnet_weights(rs, stacks, stem=stem, bias_std=0.2)
    L, D = 16, 32
    w.update(gridtd_weights(rs, L, D, H, H, V))
    X = rs.uniform(-120, 130, size=(1, hw, hw, 3)).astype(np.float32)
    cap = [7, 19, 33, 1]
    host, _, _ = _engine(stacks, stem, hw, 1, 4, w, decoder="gridtd", H=H, V=V)
    dev, _, _ = _engine(stacks, stem, hw, 1, 4, {}, decoder="gridtd", H=H, V=V)
    dev.set_weights_from_device({k: torch.as_tensor(v).cuda() for k, v in w.items()})
    outs = []
    for eng in (host, dev):
        eng.encode_images(X)
        eng.decoder_forward([cap])
        outs.append([t.clone() for t in eng.explain_tokens([0, 0, 0], [1, 2, 3], want_R_feat=True)[:2]])
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("stacks,stem,hw", [(((8, 2), (16, 3), (32, 2)), 16, 64), (((32, 2), (64, 2)), 32, 64)])
def test_resnet_weights_set_from_device_match_host_set(stacks, stem, hw):
    """lrp_set_weight_dev for the ResNet encoder (the multi-GPU start-up path: conv kernels, the 7x7 stem's three matrices,
    BatchNorm vectors packed by device kernels — resnet_encoder.h pack_unit_dev; no device-to-host copy): features and
    heat-maps bit-identical to the host-set handle, in both arithmetic modes; a second device set replaces the operands.
    (The second geometry has widths % 32 == 0: the interleaved fp16-pair dual matrices.)"""
    import torch
    rs = np.random.RandomState(8)
    w = resnet_weights(rs, stacks, stem=stem, bias_std=0.2)
    H, V = 32, 50
    side = hw // 4 // (2 ** (len(stacks) - 1))
    w.update(gridtd_weights(rs, side * side, 4 * stacks[-1][0], H, H, V))
    X = rs.uniform(-120, 130, size=(2, hw, hw, 3)).astype(np.float32)
    host, side, D = _engine(stacks, stem, hw, 2, 4, w)
    from lrp_imagecaptioning_amd.engine import LRPEngine
    dev = LRPEngine(decoder="gridtd", img_hw=(hw, hw), L=side * side, D=D, H=H, E=H, V=V, max_images=2, max_tokens=4,
                    max_caption_len=6, resnet={"stem": stem, "stacks": stacks})
    dev.set_weights_from_device({k: torch.as_tensor(v).cuda() for k, v in w.items()})
    R = rs.standard_normal((4, side * side, D)).astype(np.float32)

    def run(eng, prec):
        eng.set_precision(prec)
        eng.encode_images(X)
        return eng.get_features().clone(), eng.cnn_explain([0, 1, 1, 0], R).clone()
    for prec in ("bf16x3", "fp32"):
        fh, oh = run(host, prec)
        fd, od = run(dev, prec)
        assert torch.equal(fd, fh) and torch.equal(od, oh), prec
    w2 = {k: (v * 1.25).astype(np.float32) if k.endswith("_conv_W") else v for k, v in w.items()}
    dev.set_weights_from_device({k: torch.as_tensor(v).cuda() for k, v in w2.items() if k.endswith("_conv_W")})
    host.set_weights({k: v for k, v in w2.items() if k.endswith("_conv_W")})
    fh, oh = run(host, "bf16x3")
    fd, od = run(dev, "bf16x3")
    assert torch.equal(fd, fh) and torch.equal(od, oh) and not torch.equal(od, run(host, "fp32")[1])
