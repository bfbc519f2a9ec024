// resnet_encoder.h — host-side orchestration of the CNN half for the ResNet-101 encoder (BASELINE config 4:
// grid-TD decoder + ResNet-101 cut at conv5_block3_out, LRP-alpha1beta0).  Same contract as Encoder (encoder.h):
//   encode():  one forward per IMAGE, caching per conv unit the gate that combines the BatchNorm reverse rule
//              (RA:197-257) with the alpha1beta0 denominator (RR:274-322) and per block the Add-rule factors (RA:260-286)
//   explain(): one reverse walk per TOKEN: three (four) convs per bottleneck block on conv_igemm + streaming kernels
// The algorithm is oracle/resnet_lrp_ref.analyze_cached, which equals the literal iNNvestigate walk to 1e-10.
// This path runs in exact fp32 (PREC_FP32) throughout.
#pragma once
#include <algorithm>
#include <string>
#include <vector>

#include "common.h"
#include "conv_igemm.h"
#include "f16_operand.h"
#include "resnet_kernels.h"

namespace lrp {

constexpr float RN_BN_EPS = 1.001e-5f;

struct RnUnit {            // conv + BN
  std::string name;
  int k = 1, cin = 0, cout = 0, stride = 1;
  int Hin = 0, Win = 0, Hout = 0, Wout = 0;
  bool relu = false;
  bool have[6] = {false, false, false, false, false, false};   // W b gamma beta mean var
  DevBuf w_a, w_z, w_b, w_bs, bias, gamma, beta, mean, var;   // w_bs: w_b in split8 form (bf16x3 reverse walk)
  DevBuf w_dual;   // [w rows | w+ rows]: c and Z+ of the unit in one conv pass (EPI_FWD_DUAL without the relu)
  DevBuf w_dual_h, wds;    // w_dual as fp16 pairs scaled by a power of two + its scale record (f16_operand.h)
  bool dual_il = false;    // ... with its rows interleaved per 32 channels ([w | w+] side by side): BN + gate in the conv's epilogue
  DevBuf gate;             // [B][Hout][Wout][cout]: act*Q (relu units) or Q (pre-Add units)
  size_t out_elems() const { return (size_t)Hout * Wout * cout; }
  size_t in_elems() const { return (size_t)Hin * Win * cin; }
};

struct RnBlock {
  int u0 = -1, u1 = -1, u2 = -1, u3 = -1;     // unit indices (u0 = projection shortcut or -1)
  int stride = 1, cin = 0, f = 0, Hin = 0, Win = 0, H = 0, W = 0;
  DevBuf t_in;             // [B][Hin][Win][cin] block input (what relevance is multiplied with at the fork)
  DevBuf t_sub;            // [B][H][W][cin] stride-2 gather of t_in (first block of stacks 3..5)
  DevBuf GA, GS;           // [B][H][W][4f]: fA*Q3 and fS (identity) / fS*Q0 (projection)
};

struct ResNetEncoder {
  int img_h = 0, img_w = 0, max_images = 0, max_tokens = 0, stem_c = 0;
  int top_h = 0, top_w = 0, top_c = 0;
  std::vector<RnUnit> units;       // units[0] = stem
  std::vector<RnBlock> blocks;
  DevBuf images, stemA, a0;        // image copy, stem im2col, stem activation [B][H/2][W/2][stem]
  DevBuf fa, fb, fc, fz, fsc;      // forward scratch
  DevBuf fxs;                      // a unit's input as scaled fp16 pairs (fp16-pair forward)
  DevBuf f16_slots;                // scratch maxima of make_f16_operand
  DevBuf act_max, act_unscale;     // ACT_MAX_SLOTS maxima per unit output / block output; 2^-k per unit input
  // round-4 forward (encode_emit): per IMAGE max-slots of every unit / block output, the pair scale of every tensor that is
  // written as pairs, the unscale of every unit's input, and per unit the bound constants {A, B} (rn_unit_norm_kernel)
  DevBuf eslots, eoscale, eunscale, fnorm;
  bool norms_ready = false;
  DevBuf feat;                     // [B][top...]
  DevBuf r0, r1, r2, r3, r4, r5;   // reverse scratch (per token)
  int encoded = 0;
  bool features_only = false;
  bool profile = false;
  int prec = PREC_BF16X3;  // arithmetic of the reverse walk's conv chains (lrp_set_precision); forward stays exact fp32
  std::vector<ProfileRec> prof;

  int add_unit(const std::string& nm, int k, int cin, int cout, int stride, int Hin, int Win, bool relu) {
    RnUnit u;
    u.name = nm; u.k = k; u.cin = cin; u.cout = cout; u.stride = stride; u.Hin = Hin; u.Win = Win; u.relu = relu;
    u.Hout = (Hin + stride - 1) / stride; u.Wout = (Win + stride - 1) / stride;
    units.push_back(std::move(u));
    return (int)units.size() - 1;
  }

  int init(const lrp_config& c, int64_t* total) {
    img_h = c.img_h; img_w = c.img_w; max_images = c.max_images; max_tokens = c.max_tokens; stem_c = c.resnet_stem;
    if (c.resnet_n_stacks < 1 || c.resnet_n_stacks > 8) return fail(LRP_ERR_INVALID, "resnet_n_stacks out of range");
    if ((img_h % 4) || (img_w % 4) || stem_c % 4) return fail(LRP_ERR_UNSUPPORTED, "image size and stem width must be multiples of 4");
    add_unit("conv1", 7, 3, stem_c, 2, img_h, img_w, true);
    int H = img_h / 4, W = img_w / 4, cin = stem_c;
    for (int s = 0; s < c.resnet_n_stacks; ++s) {
      const int f = c.resnet_filters[s], nb = c.resnet_blocks[s];
      if (f % 4 || nb < 1) return fail(LRP_ERR_INVALID, "bad resnet stack %d", s);
      for (int b = 1; b <= nb; ++b) {
        char nm[64];
        snprintf(nm, sizeof(nm), "conv%d_block%d", s + 2, b);
        const int stride = (b == 1 && s > 0) ? 2 : 1;
        if (stride == 2 && ((H & 1) || (W & 1))) return fail(LRP_ERR_UNSUPPORTED, "odd resolution before a stride-2 block");
        RnBlock B;
        B.stride = stride; B.cin = cin; B.f = f; B.Hin = H; B.Win = W; B.H = H / stride; B.W = W / stride;
        if (b == 1) B.u0 = add_unit(std::string(nm) + "_0", 1, cin, 4 * f, stride, H, W, false);
        B.u1 = add_unit(std::string(nm) + "_1", 1, cin, f, stride, H, W, true);
        B.u2 = add_unit(std::string(nm) + "_2", 3, f, f, 1, B.H, B.W, true);
        B.u3 = add_unit(std::string(nm) + "_3", 1, f, 4 * f, 1, B.H, B.W, false);
        blocks.push_back(std::move(B));
        H /= stride; W /= stride; cin = 4 * f;
      }
    }
    top_h = H; top_w = W; top_c = cin;
    if (top_h * top_w != c.L || top_c != c.D)
      return fail(LRP_ERR_INVALID, "ResNet output (%d x %d x %d) does not match L=%d, D=%d", top_h, top_w, top_c, c.L, c.D);
    const size_t B = max_images, NT = max_tokens;
    size_t max_act = (size_t)(img_h / 2) * (img_w / 2) * stem_c;
    for (const RnUnit& u : units) { max_act = std::max(max_act, u.out_elems()); max_act = std::max(max_act, u.in_elems()); }
    const size_t stem_hw = (size_t)(img_h / 2) * (img_w / 2);
    LRP_TRY(images.alloc(B * img_h * img_w * 3 * 4, total));
    LRP_TRY(stemA.alloc(B * stem_hw * 2 * RN_STEM_K * 4, total));
    LRP_TRY(a0.alloc(B * stem_hw * stem_c * 4, total));
    LRP_TRY(q_stem.alloc(B * stem_hw * stem_c * 4, total));
    LRP_TRY(pool_win.alloc(B * (stem_hw / 4) * stem_c, total));      // winner of every 3x3/2 pool window (one byte)
    for (DevBuf* d : {&fa, &fb, &fc, &fz, &fsc, &fxs}) LRP_TRY(d->alloc(B * max_act * 4, total));
    LRP_TRY(act_max.alloc((units.size() + blocks.size()) * ACT_MAX_SLOTS * sizeof(unsigned), total));
    LRP_TRY(act_unscale.alloc(units.size() * sizeof(float), total));
    LRP_TRY(eslots.alloc((units.size() + blocks.size()) * B * ACT_MAX_SLOTS * sizeof(unsigned), total));
    LRP_TRY(eoscale.alloc(units.size() * B * sizeof(float), total));
    LRP_TRY(eunscale.alloc(units.size() * B * sizeof(float), total));
    LRP_TRY(fnorm.alloc(units.size() * 2 * sizeof(float), total));
    LRP_TRY(feat.alloc(B * (size_t)top_h * top_w * top_c * 4, total));
    const size_t max_tok = std::max(max_act, stem_hw * (size_t)std::max(RN_STEM_TCOLS, stem_c));
    for (DevBuf* d : {&r0, &r1, &r2, &r3, &r4, &r5}) LRP_TRY(d->alloc(NT * max_tok * 4, total));
    for (RnUnit& u : units) LRP_TRY(u.gate.alloc(B * u.out_elems() * 4, total));
    for (RnBlock& b : blocks) {
      LRP_TRY(b.t_in.alloc(B * (size_t)b.Hin * b.Win * b.cin * 4, total));
      if (b.stride == 2) LRP_TRY(b.t_sub.alloc(B * (size_t)b.H * b.W * b.cin * 4, total));
      LRP_TRY(b.GA.alloc(B * (size_t)b.H * b.W * 4 * b.f * 4, total));
      LRP_TRY(b.GS.alloc(B * (size_t)b.H * b.W * 4 * b.f * 4, total));
    }
    // staging of lrp_set_weight_dev (allocated here: that entry point promises no allocation / stream synchronisation)
    size_t mx = 0, mxd = 0;
    for (const RnUnit& q : units) {
      mx = std::max(mx, (size_t)q.k * q.k * q.cin * q.cout);
      mxd = std::max(mxd, (size_t)conv_npad(2 * q.cout) * q.k * q.k * conv_cinp(q.cin));
    }
    LRP_TRY(raw_tmp.alloc(mx * 4, total));
    LRP_TRY(pack_tmp.alloc(mxd * 4, total));
    return LRP_OK;
  }
  ~ResNetEncoder() {
    if (ev_pack) (void)hipEventDestroy(ev_pack);
  }

  int find_unit(const std::string& nm) const {
    for (size_t i = 0; i < units.size(); ++i)
      if (units[i].name == nm) return (int)i;
    return -1;
  }

  static int up(DevBuf& d, const std::vector<float>& v, int64_t* total) {
    LRP_TRY(d.alloc(v.size() * 4, total));
    LRP_HIP_CHECK(hipMemcpy(d.p, v.data(), v.size() * 4, hipMemcpyHostToDevice));
    return LRP_OK;
  }

  // name = "<unit>_conv_W" | "_conv_b" | "_bn_gamma" | "_bn_beta" | "_bn_mean" | "_bn_var"; returns 1 if not ours
  int set_weight(const std::string& nm, const float* data, int ndim, const int64_t* shape, int64_t* total) {
    static const char* suf[6] = {"_conv_W", "_conv_b", "_bn_gamma", "_bn_beta", "_bn_mean", "_bn_var"};
    for (int s = 0; s < 6; ++s) {
      const std::string sf(suf[s]);
      if (nm.size() <= sf.size() || nm.compare(nm.size() - sf.size(), sf.size(), sf) != 0) continue;
      const int ui = find_unit(nm.substr(0, nm.size() - sf.size()));
      if (ui < 0) return 1;
      RnUnit& u = units[ui];
      if (s == 0) {
        if (ndim != 4 || shape[0] != u.k || shape[1] != u.k || shape[2] != u.cin || shape[3] != u.cout)
          return fail(LRP_ERR_INVALID, "%s: expected HWIO (%d,%d,%d,%d)", nm.c_str(), u.k, u.k, u.cin, u.cout);
        LRP_TRY(pack_unit(u, data, total));
      } else {
        if (ndim != 1 || shape[0] != u.cout) return fail(LRP_ERR_INVALID, "%s: expected (%d,)", nm.c_str(), u.cout);
        DevBuf* dst[6] = {nullptr, &u.bias, &u.gamma, &u.beta, &u.mean, &u.var};
        LRP_TRY(up(*dst[s], std::vector<float>(data, data + u.cout), total));
      }
      u.have[s] = true;
      norms_ready = false;
      return LRP_OK;
    }
    return 1;
  }

  // lrp_set_weight_dev for the encoder units: the array is already in HBM (RCCL broadcast); vectors are D2D copies, kernels
  // are packed by device kernels — no device-to-host copy, no stream synchronisation.  Returns 1 if the name is not ours.
  // ONE staging pair for every unit (largest interleaved dual matrix / largest HWIO kernel): a call's packers read it
  // asynchronously on that call's stream, so the NEXT call — possibly on another stream: an RCCL stream, then a compute
  // stream — first makes its stream wait for `ev_pack`, recorded behind the previous call's last reader.
  DevBuf pack_tmp, raw_tmp;
  hipEvent_t ev_pack = nullptr;
  int set_weight_dev(const std::string& nm, const float* data_dev, int ndim, const int64_t* shape, int64_t* total, hipStream_t st) {
    static const char* suf[6] = {"_conv_W", "_conv_b", "_bn_gamma", "_bn_beta", "_bn_mean", "_bn_var"};
    for (int s = 0; s < 6; ++s) {
      const std::string sf(suf[s]);
      if (nm.size() <= sf.size() || nm.compare(nm.size() - sf.size(), sf.size(), sf) != 0) continue;
      const int ui = find_unit(nm.substr(0, nm.size() - sf.size()));
      if (ui < 0) return 1;
      RnUnit& u = units[ui];
      if (s == 0) {
        if (ndim != 4 || shape[0] != u.k || shape[1] != u.k || shape[2] != u.cin || shape[3] != u.cout)
          return fail(LRP_ERR_INVALID, "%s: expected HWIO (%d,%d,%d,%d)", nm.c_str(), u.k, u.k, u.cin, u.cout);
        // own copy first: the packers run asynchronously, the caller's buffer need not outlive this call
        const size_t nW = (size_t)u.k * u.k * u.cin * u.cout;
        if (!ev_pack) LRP_HIP_CHECK(hipEventCreateWithFlags(&ev_pack, hipEventDisableTiming));
        else LRP_HIP_CHECK(hipStreamWaitEvent(st, ev_pack, 0));          // the previous unit's packers still read the staging pair
        LRP_HIP_CHECK(hipMemcpyAsync(raw_tmp.p, data_dev, nW * 4, hipMemcpyDeviceToDevice, st));
        LRP_TRY(pack_unit_dev(u, raw_tmp.as<float>(), total, st));
        LRP_HIP_CHECK(hipEventRecord(ev_pack, st));
      } else {
        if (ndim != 1 || shape[0] != u.cout) return fail(LRP_ERR_INVALID, "%s: expected (%d,)", nm.c_str(), u.cout);
        DevBuf* dst[6] = {nullptr, &u.bias, &u.gamma, &u.beta, &u.mean, &u.var};
        if (!dst[s]->p || dst[s]->bytes != (size_t)u.cout * 4) LRP_TRY(dst[s]->alloc((size_t)u.cout * 4, total));
        LRP_HIP_CHECK(hipMemcpyAsync(dst[s]->p, data_dev, (size_t)u.cout * 4, hipMemcpyDeviceToDevice, st));
      }
      u.have[s] = true;
      norms_ready = false;
      encoded = 0;                                       // caches belong to the old weights
      return LRP_OK;
    }
    return 1;
  }
  int pack_unit_dev(RnUnit& u, const float* w_dev, int64_t* total, hipStream_t st) {
    auto mk = [&](DevBuf& d, size_t floats) -> int {
      if (d.p && d.bytes == floats * 4) return LRP_OK;
      return d.alloc(floats * 4, total);
    };
    auto split = [&](const float* src, float* dst, size_t n) {
      hipLaunchKernelGGL(split_copy_kernel, dim3(stream_grid(n / 8)), dim3(256), 0, st, src, dst, n / 8);
    };
    if (u.k == 7) {
      const int Np = conv_npad(u.cout), K = 2 * RN_STEM_K, Npb = conv_npad(RN_STEM_TCOLS), Kb = conv_cinp(u.cout);
      LRP_TRY(mk(u.w_a, (size_t)Np * K)); LRP_TRY(mk(u.w_z, (size_t)Np * K)); LRP_TRY(mk(u.w_b, (size_t)Npb * Kb));
      hipLaunchKernelGGL(rn_pack_stem_dev_kernel, dim3(stream_grid((size_t)Np * K + (size_t)Npb * Kb)), dim3(256), 0, st, w_dev,
                         u.w_a.as<float>(), u.w_z.as<float>(), u.w_b.as<float>(), u.cout, Np, Npb, Kb);
      if (!(u.cout & 7)) {
        LRP_TRY(mk(u.w_bs, (size_t)Npb * Kb));
        split(u.w_b.as<float>(), u.w_bs.as<float>(), (size_t)Npb * Kb);
      }
      LRP_HIP_CHECK(hipGetLastError());
      return LRP_OK;
    }
    const int taps = u.k * u.k, CPi = conv_cinp(u.cin), CPo = conv_cinp(u.cout);
    const int Np = conv_npad(u.cout), Npb = conv_npad(u.cin);
    const size_t nf = (size_t)Np * taps * CPi, nb = (size_t)Npb * taps * CPo;
    auto pack = [&](float* dst, int bwd, int rows, int dual, int pos) {
      const size_t tot = (size_t)rows * taps * (bwd ? CPo : CPi);
      hipLaunchKernelGGL(pack_conv_dev_kernel, dim3(stream_grid(tot)), dim3(256), 0, st, w_dev, dst, bwd, u.cin, u.cout, bwd ? CPo : CPi,
                         rows, dual, pos, taps);
    };
    LRP_TRY(mk(u.w_a, nf)); LRP_TRY(mk(u.w_z, nf));
    pack(u.w_a.as<float>(), 0, Np, 0, 0);
    pack(u.w_z.as<float>(), 0, Np, 0, 1);                // inputs are post-ReLU: Z = conv(x, w+) + b
    if (!(u.cout & 3)) {
      const int Nd = conv_npad(2 * u.cout);
      const size_t nd = (size_t)Nd * taps * CPi;
      LRP_TRY(mk(u.w_dual, nd));
      pack(u.w_dual.as<float>(), 0, Nd, 1, 0);
      if (!(u.cin & 7)) {
        u.dual_il = !(u.cout & 31) && Nd == 2 * u.cout;
        const float* src = u.w_dual.as<float>();
        if (u.dual_il) {                                  // rows in blocks of 32: [w | w+] of the same 32 channels
          pack(pack_tmp.as<float>(), 0, Nd, 2, 0);
          src = pack_tmp.as<float>();
        }
        LRP_TRY(make_f16_operand(f16_slots, src, nd, 0, 0, u.w_dual_h, u.wds, total, st, false));
      }
    }
    LRP_TRY(mk(u.w_b, nb)); LRP_TRY(mk(u.w_bs, nb));
    pack(u.w_b.as<float>(), 1, Npb, 0, 1);
    split(u.w_b.as<float>(), u.w_bs.as<float>(), nb);
    LRP_HIP_CHECK(hipGetLastError());
    return LRP_OK;
  }

  int pack_unit(RnUnit& u, const float* w, int64_t* total) {
    const size_t nW = (size_t)u.k * u.k * u.cin * u.cout;
    std::vector<float> wp(nW), wn(nW), pk;
    for (size_t i = 0; i < nW; ++i) { wp[i] = w[i] >= 0.f ? w[i] : 0.f; wn[i] = w[i] < 0.f ? w[i] : 0.f; }
    if (u.k == 7) {
      // stem forward: 1-tap GEMM over the im2col matrix [x+ patch | x- patch] (K = 2 * 160)
      const int Np = conv_npad(u.cout), K = 2 * RN_STEM_K;
      std::vector<float> a((size_t)Np * K, 0.f), z((size_t)Np * K, 0.f);
      for (int kk = 0; kk < 147; ++kk)
        for (int co = 0; co < u.cout; ++co) {
          const size_t s = (size_t)kk * u.cout + co;
          a[(size_t)co * K + kk] = w[s]; a[(size_t)co * K + RN_STEM_K + kk] = w[s];
          z[(size_t)co * K + kk] = wp[s]; z[(size_t)co * K + RN_STEM_K + kk] = wn[s];
        }
      LRP_TRY(up(u.w_a, a, total));
      LRP_TRY(up(u.w_z, z, total));
      // reverse at the image: T[q][tap*6 + c] (c<3: w+, c>=3: w-), K = cout
      const int Npb = conv_npad(RN_STEM_TCOLS), Kb = conv_cinp(u.cout);
      pk.assign((size_t)Npb * Kb, 0.f);
      for (int t = 0; t < 49; ++t)
        for (int c = 0; c < 3; ++c)
          for (int co = 0; co < u.cout; ++co) {
            pk[(size_t)(t * 6 + c) * Kb + co] = wp[((size_t)t * 3 + c) * u.cout + co];
            pk[(size_t)(t * 6 + 3 + c) * Kb + co] = wn[((size_t)t * 3 + c) * u.cout + co];
          }
      LRP_TRY(up(u.w_b, pk, total));
      if (!(u.cout & 7)) {                               // split-bf16 copy: the tap GEMM of the walk as bf16x3
        std::vector<float> sp(pk.size());
        pack_split8(pk.data(), pk.size(), sp.data());
        LRP_TRY(up(u.w_bs, sp, total));
      }
      return LRP_OK;
    }
    const int taps = u.k * u.k;
    const int Np = conv_npad(u.cout), K = taps * conv_cinp(u.cin);
    pk.assign((size_t)Np * K, 0.f);
    pack_conv_fwd(w, taps, u.cin, u.cout, 0, Np, pk.data());
    LRP_TRY(up(u.w_a, pk, total));
    pk.assign((size_t)Np * K, 0.f);
    pack_conv_fwd(wp.data(), taps, u.cin, u.cout, 0, Np, pk.data());           // inputs are post-ReLU: Z = conv(x, w+) + b
    LRP_TRY(up(u.w_z, pk, total));
    if (!(u.cout & 3)) {
      const int Nd = conv_npad(2 * u.cout);
      pk.assign((size_t)Nd * K, 0.f);
      pack_conv_fwd(w, taps, u.cin, u.cout, 0, Nd, pk.data());
      pack_conv_fwd(wp.data(), taps, u.cin, u.cout, u.cout, Nd, pk.data());
      LRP_TRY(up(u.w_dual, pk, total));
      if (!(u.cin & 7)) {
        u.dual_il = !(u.cout & 31) && Nd == 2 * u.cout;
        if (u.dual_il) {                                  // rows in blocks of 32: [w | w+] of the same 32 channels
          std::vector<float> il(pk.size(), 0.f);
          for (int c = 0; c < u.cout; ++c)
            for (int half = 0; half < 2; ++half)
              memcpy(&il[(size_t)(64 * (c / 32) + 32 * half + (c & 31)) * K], &pk[(size_t)(half * u.cout + c) * K], (size_t)K * sizeof(float));
          DevBuf tmp;
          LRP_TRY(up(tmp, il, nullptr));
          LRP_TRY(make_f16_operand(f16_slots, tmp.as<float>(), pk.size(), 0, 0, u.w_dual_h, u.wds, total, nullptr));
        } else {
          LRP_TRY(make_f16_operand(f16_slots, u.w_dual.as<float>(), pk.size(), 0, 0, u.w_dual_h, u.wds, total, nullptr));
        }
      }
    }
    const int Npb = conv_npad(u.cin), Kb = taps * conv_cinp(u.cout);
    pk.assign((size_t)Npb * Kb, 0.f);
    pack_conv_bwd(wp.data(), taps, u.cin, u.cout, 0, pk.data());
    LRP_TRY(up(u.w_b, pk, total));
    std::vector<float> sp(pk.size());
    pack_split8(pk.data(), pk.size(), sp.data());
    return up(u.w_bs, sp, total);
  }

  int check_ready() const {
    for (const RnUnit& u : units)
      for (int s = 0; s < 6; ++s)
        if (!u.have[s]) return fail(LRP_ERR_STATE, "ResNet weights of '%s' incomplete", u.name.c_str());
    return LRP_OK;
  }

  unsigned* unit_slots(int ui) { return act_max.as<unsigned>() + (size_t)ui * ACT_MAX_SLOTS; }
  unsigned* block_slots(size_t bi) { return act_max.as<unsigned>() + (units.size() + bi) * ACT_MAX_SLOTS; }

  // conv + BN unit forward: x [B][Hin][Win][cin] -> act (relu(BN) or BN) and the unit's gate.
  // slots_in: maxima of x (fp16-pair forward: the power of two x is scaled by); slots_out: where max|act| is collected
  // (nullptr: nobody convolves this output)
  int unit_forward(RnUnit& u, const float* x, int B, float* act, hipStream_t st, const unsigned* slots_in = nullptr,
                   unsigned* slots_out = nullptr) {
    const float* xin = x;
    if (u.k == 1 && u.stride == 2) {
      const size_t n = (size_t)B * u.Hout * u.Wout * u.cin;
      hipLaunchKernelGGL(rn_subsample2_kernel, dim3(stream_grid(n)), dim3(256), 0, st, x, fsc.as<float>(), B, u.Hin, u.Win, u.cin);
      LRP_HIP_CHECK(hipGetLastError());
      xin = fsc.as<float>();
    }
    ConvArgs ca{};
    ca.in = xin; ca.bias = u.bias.as<float>(); ca.N = u.cout;
    if (u.k == 3) { ca.NB = B; ca.H = u.Hout; ca.W = u.Wout; ca.Cin = u.cin; ca.CinP = conv_cinp(u.cin); ca.taps = 9; }
    else { ca.NB = B * u.Hout * u.Wout; ca.H = 1; ca.W = 1; ca.Cin = u.cin; ca.CinP = conv_cinp(u.cin); ca.taps = 1; }
    ConvArgs cz = ca;
    const int ui = (int)(&u - units.data());
    if (u.w_dual_h.p && slots_in && prec == PREC_BF16X3) {
      // the same single pass on the fp16 MFMA: x and (w | w+) as fp16 pairs, the product in three MFMAs with blocked
      // fp32 accumulation (conv_igemm.h PREC_F16X2) — fp32-grade, ~2.5x the fp32 matrix rate
      const size_t n8 = (size_t)B * u.Hout * u.Wout * u.cin / 8;
      hipLaunchKernelGGL(split_h_scaled_kernel, dim3(stream_grid(n8)), dim3(256), 0, st, xin, fxs.as<float>(), n8, slots_in,
                         act_unscale.as<float>() + ui, u.wds.as<float>());
      LRP_HIP_CHECK(hipGetLastError());
      ca.in = fxs.as<float>(); ca.wpk = u.w_dual_h.as<float>(); ca.N = 2 * u.cout; ca.split = u.cout; ca.dual_norelu = 1;
      ca.out = fc.as<float>(); ca.out2 = fz.as<float>(); ca.in_unscale = act_unscale.as<float>() + ui;
      if (u.dual_il) {
        // BatchNorm, ReLU and the unit's gate in the conv's epilogue: c and Z+ never go to memory (conv_igemm.h dual_gate = 2)
        ca.dual_il = 1; ca.dual_gate = 2; ca.bn_gamma = u.gamma.as<float>(); ca.bn_beta = u.beta.as<float>();
        ca.bn_mean = u.mean.as<float>(); ca.bn_var = u.var.as<float>(); ca.bn_eps = RN_BN_EPS; ca.bn_relu = u.relu ? 1 : 0;
        ca.out = act; ca.out2 = u.gate.as<float>(); ca.act_max_out = slots_out;
        LRP_HIP_CHECK(conv_launch(EPI_FWD_DUAL, ca, st, PREC_F16X2));
        return LRP_OK;
      }
      LRP_HIP_CHECK(conv_launch(EPI_FWD_DUAL, ca, st, PREC_F16X2));
    } else if (u.w_dual.p) {
      // c = conv(x, w) + b and Z = conv(x, w+) + b in ONE pass over x: the A tile is staged once for both
      ca.wpk = u.w_dual.as<float>(); ca.N = 2 * u.cout; ca.split = u.cout; ca.dual_norelu = 1;
      ca.out = fc.as<float>(); ca.out2 = fz.as<float>();
      LRP_HIP_CHECK(conv_launch(EPI_FWD_DUAL, ca, st));
    } else {
      ca.wpk = u.w_a.as<float>(); ca.out = fc.as<float>();
      LRP_HIP_CHECK(conv_launch(EPI_BIAS, ca, st));
      cz.wpk = u.w_z.as<float>(); cz.out = fz.as<float>();
      LRP_HIP_CHECK(conv_launch(EPI_BIAS, cz, st));
    }
    const size_t n = (size_t)B * u.out_elems();
    hipLaunchKernelGGL(rn_bn_unit_kernel, dim3(stream_grid(n)), dim3(256), 0, st, fc.as<float>(), fz.as<float>(),
                       u.gamma.as<float>(), u.beta.as<float>(), u.mean.as<float>(), u.var.as<float>(), RN_BN_EPS, act,
                       u.gate.as<float>(), (float*)nullptr, n, u.cout, u.relu ? 1 : 0, slots_out);
    LRP_HIP_CHECK(hipGetLastError());
    return LRP_OK;
  }

  // ---- round 4: no pass between two convs (kernels and the bounds behind the scales: resnet_kernels.h) ----
  unsigned* eslots_unit(int ui) { return eslots.as<unsigned>() + (size_t)ui * max_images * ACT_MAX_SLOTS; }
  unsigned* eslots_block(size_t bi) { return eslots.as<unsigned>() + (units.size() + bi) * max_images * ACT_MAX_SLOTS; }
  float* eoscale_unit(int ui) { return eoscale.as<float>() + (size_t)ui * max_images; }
  float* eunscale_unit(int ui) { return eunscale.as<float>() + (size_t)ui * max_images; }
  float* fnorm_unit(int ui) { return fnorm.as<float>() + (size_t)ui * 2; }
  static dim3 img_grid(size_t per_img_items, int B) {
    size_t gx = (per_img_items + 255) / 256, cap = (size_t)std::max(1, 256 * 8 / B);
    return dim3((unsigned)std::max<size_t>(1, std::min(gx, cap)), (unsigned)B);
  }
  // the whole network on the pair-emitting path: fp16-pair forward, every unit behind the stem with interleaved dual rows
  // (cout % 32 == 0) and whole split8 groups on its input (cin % 8 == 0).  ResNet-50/101/152 qualify; LRP_FWD_EMIT=0 = round 3's path
  bool emit_ok() const {
    if (prec != PREC_BF16X3 || !sw().fwd_emit || (stem_c & 7)) return false;
    for (size_t i = 1; i < units.size(); ++i)
      if (!units[i].w_dual_h.p || !units[i].dual_il || (units[i].cin & 7)) return false;
    return true;
  }
  int unit_norms(hipStream_t st) {
    LRP_HIP_CHECK(hipMemsetAsync(fnorm.p, 0, fnorm.bytes, st));
    for (size_t i = 1; i < units.size(); ++i) {
      const RnUnit& u = units[i];
      hipLaunchKernelGGL(rn_unit_norm_kernel, dim3(u.cout), dim3(256), 0, st, u.w_a.as<float>(), u.k * u.k * conv_cinp(u.cin),
                         u.bias.as<float>(), u.gamma.as<float>(), u.beta.as<float>(), u.mean.as<float>(), u.var.as<float>(),
                         RN_BN_EPS, fnorm_unit((int)i));
    }
    LRP_HIP_CHECK(hipGetLastError());
    norms_ready = true;
    return LRP_OK;
  }
  // conv + BN unit on pairs: x (pairs, scaled per image) -> gate, max-slots and EITHER the activation as the next conv's pairs
  // (pairs_out: relu units inside a block; fp32 activation not written) OR the fp32 pre-Add tensor `act`
  int unit_forward_emit(RnUnit& u, const float* xpairs, int B, float* act, float* pairs_out, hipStream_t st) {
    const int ui = (int)(&u - units.data());
    ConvArgs ca{};
    ca.in = xpairs; ca.bias = u.bias.as<float>();
    if (u.k == 3) { ca.NB = B; ca.H = u.Hout; ca.W = u.Wout; ca.taps = 9; }
    else { ca.NB = B * u.Hout * u.Wout; ca.H = 1; ca.W = 1; ca.taps = 1; }
    ca.Cin = u.cin; ca.CinP = conv_cinp(u.cin);
    ca.wpk = u.w_dual_h.as<float>(); ca.N = 2 * u.cout; ca.split = u.cout; ca.dual_norelu = 1;
    ca.in_unscale = eunscale_unit(ui);
    ca.dual_il = 1; ca.dual_gate = 2; ca.bn_gamma = u.gamma.as<float>(); ca.bn_beta = u.beta.as<float>();
    ca.bn_mean = u.mean.as<float>(); ca.bn_var = u.var.as<float>(); ca.bn_eps = RN_BN_EPS; ca.bn_relu = u.relu ? 1 : 0;
    ca.out = act; ca.out2 = u.gate.as<float>();
    ca.act_max_out = pairs_out ? (unsigned*)nullptr : eslots_unit(ui);      // measured maxima: the block end's two summands only
    ca.scale_per_img = 1; ca.img_rows = u.Hout * u.Wout; ca.n_imgs = B;
    if (pairs_out) { ca.pairs_out = pairs_out; ca.pairs_scale = eoscale_unit(ui); ca.skip_out = 1; }
    LRP_HIP_CHECK(conv_launch(EPI_FWD_DUAL, ca, st, PREC_F16X2));
    return LRP_OK;
  }
  int encode_emit(int B, hipStream_t st) {
    if (!norms_ready) LRP_TRY(unit_norms(st));
    LRP_HIP_CHECK(hipMemsetAsync(eslots.p, 0, eslots.bytes, st));
    RnUnit& s = units[0];
    auto scales_of = [&](const RnBlock& rb) {             // what the producer of rb's input derives for rb (resnet_kernels.h)
      RnBlockScales S{};
      S.norm1 = fnorm_unit(rb.u1); S.norm2 = fnorm_unit(rb.u2);
      S.wsc1 = units[rb.u1].wds.as<float>(); S.wsc2 = units[rb.u2].wds.as<float>(); S.wsc3 = units[rb.u3].wds.as<float>();
      S.wsc0 = rb.u0 >= 0 ? units[rb.u0].wds.as<float>() : (const float*)nullptr;
      S.osc_t = eoscale_unit(rb.u3);                     // (slot of a tensor that is never written as pairs itself)
      S.osc1 = eoscale_unit(rb.u1); S.osc2 = eoscale_unit(rb.u2);
      S.us1 = eunscale_unit(rb.u1); S.us2 = eunscale_unit(rb.u2); S.us3 = eunscale_unit(rb.u3);
      S.us0 = rb.u0 >= 0 ? eunscale_unit(rb.u0) : (float*)nullptr;
      return S;
    };
    float* tp = fc.as<float>();                          // the block input as pairs (fc / fsc alternate)
    float* tp_other = fsc.as<float>();
    {  // stem: im2col -> two 1-tap GEMMs (fp32 MFMA) -> BN + relu + gate -> pool (+ pairs of the first block's input)
      const size_t tot = (size_t)B * s.Hout * s.Wout * 2 * RN_STEM_K;
      hipLaunchKernelGGL(rn_stem_im2col_kernel, dim3(stream_grid(tot)), dim3(256), 0, st, images.as<float>(),
                         stemA.as<float>(), B, img_h, img_w);
      LRP_HIP_CHECK(hipGetLastError());
      ConvArgs ca{};
      ca.in = stemA.as<float>(); ca.NB = B * s.Hout * s.Wout; ca.H = 1; ca.W = 1; ca.Cin = 2 * RN_STEM_K; ca.CinP = 2 * RN_STEM_K;
      ca.taps = 1; ca.N = s.cout; ca.bias = s.bias.as<float>();
      ConvArgs cz = ca;
      ca.wpk = s.w_a.as<float>(); ca.out = fa.as<float>();
      LRP_HIP_CHECK(conv_launch(EPI_BIAS, ca, st));
      cz.wpk = s.w_z.as<float>(); cz.out = fz.as<float>();
      LRP_HIP_CHECK(conv_launch(EPI_BIAS, cz, st));
      const size_t per = s.out_elems();
      hipLaunchKernelGGL(rn_bn_unit_img_kernel, img_grid(per, B), dim3(256), 0, st, fa.as<float>(), fz.as<float>(),
                         s.gamma.as<float>(), s.beta.as<float>(), s.mean.as<float>(), s.var.as<float>(), RN_BN_EPS,
                         a0.as<float>(), s.gate.as<float>(), q_stem.as<float>(), per, s.cout, 1, eslots_unit(0));
      hipLaunchKernelGGL(rn_pool3_pairs_kernel, img_grid(per / 4 / 8, B), dim3(256), 0, st, a0.as<float>(), blocks[0].t_in.as<float>(),
                         pool_win.as<unsigned char>(), tp, eslots_unit(0), scales_of(blocks[0]), s.Hout, s.Wout, s.cout);
      LRP_HIP_CHECK(hipGetLastError());
    }
    for (size_t bi = 0; bi < blocks.size(); ++bi) {
      RnBlock& b = blocks[bi];
      const float* t = b.t_in.as<float>();
      const unsigned* ts = bi == 0 ? eslots_unit(0) : eslots_block(bi - 1);      // per-image maxima of the block input
      const float* tin = tp;
      if (b.stride == 2) {
        // the walk multiplies with the fp32 sub-sampled input; the strided 1x1 convs read the same gather of the pairs
        // (a pixel's pairs are as many bytes as its fp32 row)
        const size_t n = (size_t)B * b.H * b.W * b.cin;
        hipLaunchKernelGGL(rn_subsample2_kernel, dim3(stream_grid(n)), dim3(256), 0, st, t, b.t_sub.as<float>(), B, b.Hin, b.Win, b.cin);
        hipLaunchKernelGGL(rn_subsample2_kernel, dim3(stream_grid(n)), dim3(256), 0, st, tp, fxs.as<float>(), B, b.Hin, b.Win, b.cin);
        LRP_HIP_CHECK(hipGetLastError());
        tin = fxs.as<float>();
      }
      RnUnit &u1 = units[b.u1], &u2 = units[b.u2], &u3 = units[b.u3];
      const float* sc = t;
      const unsigned* sc_slots = ts;
      if (b.u0 >= 0) {                                   // (first: it reads the gathered pairs, which unit 2 overwrites)
        LRP_TRY(unit_forward_emit(units[b.u0], tin, B, fb.as<float>(), nullptr, st));                   // fb = y0
        sc = fb.as<float>();
        sc_slots = eslots_unit(b.u0);
      }
      LRP_TRY(unit_forward_emit(u1, tin, B, nullptr, fz.as<float>(), st));                              // fz = pairs(a1)
      LRP_TRY(unit_forward_emit(u2, fz.as<float>(), B, nullptr, fxs.as<float>(), st));                  // fxs = pairs(a2)
      LRP_TRY(unit_forward_emit(u3, fxs.as<float>(), B, fa.as<float>(), nullptr, st));                  // fa = y3
      const bool last = bi + 1 == blocks.size();
      float* o = last ? feat.as<float>() : blocks[bi + 1].t_in.as<float>();
      const size_t per8 = (size_t)b.H * b.W * 4 * b.f / 8;
      hipLaunchKernelGGL(rn_block_out_pairs_kernel, img_grid(per8, B), dim3(256), 0, st, sc, fa.as<float>(), u3.gate.as<float>(),
                         b.u0 >= 0 ? units[b.u0].gate.as<float>() : (const float*)nullptr, o, b.GA.as<float>(), b.GS.as<float>(), per8,
                         last ? (unsigned*)nullptr : eslots_block(bi), last ? (float*)nullptr : tp_other, eslots_unit(b.u3), sc_slots,
                         last ? RnBlockScales{} : scales_of(blocks[bi + 1]));
      LRP_HIP_CHECK(hipGetLastError());
      std::swap(tp, tp_other);
    }
    return LRP_OK;
  }

  int encode(const float* images_dev, int B, hipStream_t st) {
    if (B < 1 || B > max_images) return fail(LRP_ERR_INVALID, "B=%d outside [1,%d]", B, max_images);
    LRP_TRY(check_ready());
    if (emit_ok()) {
      LRP_HIP_CHECK(hipMemcpyAsync(images.p, images_dev, (size_t)B * img_h * img_w * 3 * 4, hipMemcpyDeviceToDevice, st));
      LRP_TRY(encode_emit(B, st));
      encoded = B;
      features_only = false;
      return LRP_OK;
    }
    LRP_HIP_CHECK(hipMemcpyAsync(images.p, images_dev, (size_t)B * img_h * img_w * 3 * 4, hipMemcpyDeviceToDevice, st));
    LRP_HIP_CHECK(hipMemsetAsync(act_max.p, 0, act_max.bytes, st));
    RnUnit& s = units[0];
    {  // stem: im2col -> two 1-tap GEMMs (c exact / Z with both sign branches) -> BN + relu + gate
      const size_t tot = (size_t)B * s.Hout * s.Wout * 2 * RN_STEM_K;
      hipLaunchKernelGGL(rn_stem_im2col_kernel, dim3(stream_grid(tot)), dim3(256), 0, st, images.as<float>(),
                         stemA.as<float>(), B, img_h, img_w);
      LRP_HIP_CHECK(hipGetLastError());
      ConvArgs ca{};
      ca.in = stemA.as<float>(); ca.NB = B * s.Hout * s.Wout; ca.H = 1; ca.W = 1; ca.Cin = 2 * RN_STEM_K; ca.CinP = 2 * RN_STEM_K;
      ca.taps = 1; ca.N = s.cout; ca.bias = s.bias.as<float>();
      ConvArgs cz = ca;
      ca.wpk = s.w_a.as<float>(); ca.out = fc.as<float>();
      LRP_HIP_CHECK(conv_launch(EPI_BIAS, ca, st));
      cz.wpk = s.w_z.as<float>(); cz.out = fz.as<float>();
      LRP_HIP_CHECK(conv_launch(EPI_BIAS, cz, st));
      const size_t n = (size_t)B * s.out_elems();
      hipLaunchKernelGGL(rn_bn_unit_kernel, dim3(stream_grid(n)), dim3(256), 0, st, fc.as<float>(), fz.as<float>(),
                         s.gamma.as<float>(), s.beta.as<float>(), s.mean.as<float>(), s.var.as<float>(), RN_BN_EPS,
                         a0.as<float>(), s.gate.as<float>(), q_stem.as<float>(), n, s.cout, 1, unit_slots(0));
      LRP_HIP_CHECK(hipGetLastError());
      const size_t np = (size_t)B * (s.Hout / 2) * (s.Wout / 2) * s.cout;
      hipLaunchKernelGGL(rn_pool3_kernel, dim3(stream_grid(np)), dim3(256), 0, st, a0.as<float>(), blocks[0].t_in.as<float>(),
                         pool_win.as<unsigned char>(), B, s.Hout, s.Wout, s.cout);
      LRP_HIP_CHECK(hipGetLastError());
    }
    for (size_t bi = 0; bi < blocks.size(); ++bi) {
      RnBlock& b = blocks[bi];
      const float* t = b.t_in.as<float>();
      // maxima of the block input: the pooled stem activation (bounded by the stem's) or the previous block's output
      const unsigned* ts = bi == 0 ? unit_slots(0) : block_slots(bi - 1);
      if (b.stride == 2) {
        const size_t n = (size_t)B * b.H * b.W * b.cin;
        hipLaunchKernelGGL(rn_subsample2_kernel, dim3(stream_grid(n)), dim3(256), 0, st, t, b.t_sub.as<float>(), B, b.Hin,
                           b.Win, b.cin);
        LRP_HIP_CHECK(hipGetLastError());
      }
      // main path: fa <- a1, fb <- a2, fa <- y3 ; shortcut: fsc2 (= r0 scratch is per-token; use GS buffer as temp) ...
      LRP_TRY(unit_forward(units[b.u1], t, B, fa.as<float>(), st, ts, unit_slots(b.u1)));
      LRP_TRY(unit_forward(units[b.u2], fa.as<float>(), B, fb.as<float>(), st, unit_slots(b.u1), unit_slots(b.u2)));
      LRP_TRY(unit_forward(units[b.u3], fb.as<float>(), B, fa.as<float>(), st, unit_slots(b.u2)));        // fa = y3
      const float* sc = t;
      if (b.u0 >= 0) {
        LRP_TRY(unit_forward(units[b.u0], t, B, fb.as<float>(), st, ts));               // fb = y0
        sc = fb.as<float>();
      }
      const bool last = bi + 1 == blocks.size();
      float* o = last ? feat.as<float>() : blocks[bi + 1].t_in.as<float>();
      const size_t n = (size_t)B * b.H * b.W * 4 * b.f;
      hipLaunchKernelGGL(rn_block_out_kernel, dim3(stream_grid(n / 4)), dim3(256), 0, st, sc, fa.as<float>(),
                         units[b.u3].gate.as<float>(), b.u0 >= 0 ? units[b.u0].gate.as<float>() : (const float*)nullptr, o,
                         b.GA.as<float>(), b.GS.as<float>(), n, last ? (unsigned*)nullptr : block_slots(bi));
      LRP_HIP_CHECK(hipGetLastError());
    }
    encoded = B;
    features_only = false;
    return LRP_OK;
  }

  // conv-LRP step through one unit: S [n][Hout][Wout][cout] -> out [n][Hout'][..][cin] = convT(S, w+) * aux[img]
  // split: S is in split8 form and the conv runs as bf16x3; plain_out: the result feeds an element-wise kernel (fp32)
  // instead of the next conv of the chain (split8)
  struct BlockTail {                // optional: residual join and the next block's chain head, fused into the epilogue
    const float* join = nullptr; const float* join_gate = nullptr; const float* gate2 = nullptr; float* out2 = nullptr;
  };
  int unit_backward(const RnUnit& u, int n, const int* row2img, const float* S, const float* aux, float* out, hipStream_t st,
                    bool split = false, bool plain_out = true, const BlockTail* tail = nullptr) {
    ConvArgs ca{};
    if (tail) { ca.join = tail->join; ca.join_gate = tail->join_gate; ca.gate2 = tail->gate2; ca.out2s = tail->out2; }
    ca.in = S; ca.wpk = split ? u.w_bs.as<float>() : u.w_b.as<float>(); ca.row2img = row2img; ca.aux = aux; ca.out = out;
    ca.N = u.cin; ca.out_plain = plain_out ? 1 : 0;
    ca.Cin = u.cout; ca.CinP = conv_cinp(u.cout); ca.NB = n; ca.H = u.Hout; ca.W = u.Wout; ca.taps = u.k == 3 ? 9 : 1;
    ProfileRec pr{};
    if (profile) { (void)hipEventCreate(&pr.e0); (void)hipEventCreate(&pr.e1); (void)hipEventRecord(pr.e0, st); }
    LRP_HIP_CHECK(conv_launch(EPI_MUL, ca, st, split ? PREC_BF16X3 : PREC_FP32));
    if (profile) {
      (void)hipEventRecord(pr.e1, st);
      pr.flop = 2.0 * n * u.Hout * u.Wout * (double)(u.k == 3 ? 9 : 1) * u.cout * u.cin;
      prof.push_back(pr);
    }
    return LRP_OK;
  }

  int explain(int n, const int* row2img, const float* R_feat_dev, float* R_img_dev, hipStream_t st) {
    if (n < 1 || n > max_tokens) return fail(LRP_ERR_INVALID, "n=%d outside [1,%d]", n, max_tokens);
    if (encoded < 1 || features_only) return fail(LRP_ERR_STATE, "lrp_encode_images must run before the CNN explain");
    const float* Ro = R_feat_dev;          // relevance at the current block's output
    float* cur = r0.as<float>();           // where the next R_t is written (ping-pong r0 / r4)
    float* other = r4.as<float>();
    float* s3 = r1.as<float>();            // S3 = R_o * GA of the block being walked (r1 / r5 alternate)
    float* s3_other = r5.as<float>();
    bool have_s3 = false;                  // already written by the previous block's fused epilogue
    auto block_split = [&](const RnBlock& bb) {
      bool v = prec == PREC_BF16X3;
      for (int ui : {bb.u0, bb.u1, bb.u2, bb.u3})
        if (ui >= 0 && ((units[ui].cin | units[ui].cout) & 7)) v = false;
      return v;
    };
    for (int bi = (int)blocks.size() - 1; bi >= 0; --bi) {
      const RnBlock& b = blocks[bi];
      const size_t per_o = (size_t)b.H * b.W * 4 * b.f;
      if (b.u0 < 0) {
        // identity block, 3 conv launches: S3 -> S2 -> S1 -> R_t = t*C1 + R_o*GS, the join and (when the next block
        // agrees on the operand format) the next block's S3 = R_t * GA_next written by unit 1's epilogue
        const bool sp = block_split(b);
        if (!have_s3) {
          if (sp)
            hipLaunchKernelGGL(rn_mul_gate_split_kernel, dim3(stream_grid((size_t)n * per_o / 8)), dim3(256), 0, st, Ro,
                               b.GA.as<float>(), row2img, s3, n, per_o / 8);
          else
            hipLaunchKernelGGL(rn_mul_gate_kernel, dim3(stream_grid((size_t)n * per_o)), dim3(256), 0, st, Ro, b.GA.as<float>(),
                               row2img, (const float*)nullptr, s3, n, per_o);
          LRP_HIP_CHECK(hipGetLastError());
        }
        LRP_TRY(unit_backward(units[b.u3], n, row2img, s3, units[b.u2].gate.as<float>(), r2.as<float>(), st, sp, !sp));
        LRP_TRY(unit_backward(units[b.u2], n, row2img, r2.as<float>(), units[b.u1].gate.as<float>(), r3.as<float>(), st, sp, !sp));
        BlockTail tl;
        tl.join = Ro; tl.join_gate = b.GS.as<float>();
        have_s3 = false;
        if (bi > 0 && block_split(blocks[bi - 1]) == sp) {
          tl.gate2 = blocks[bi - 1].GA.as<float>(); tl.out2 = s3_other;
          have_s3 = true;
        }
        LRP_TRY(unit_backward(units[b.u1], n, row2img, r3.as<float>(), b.t_in.as<float>(), cur, st, sp, true, &tl));
        std::swap(s3, s3_other);
        Ro = cur;
        std::swap(cur, other);
        continue;
      }
      // projection block (first of a stack).  bf16x3 mode: the three convs of the main branch and the projection chain in
      // split8 form; the products that enter a chain are written split, what leaves it for the join is plain fp32
      const bool sp = block_split(b);                    // split8 groups need channel counts % 8 == 0: exact fp32 otherwise
      auto head = [&](const float* G, float* dst) {
        if (sp)
          hipLaunchKernelGGL(rn_mul_gate_split_kernel, dim3(stream_grid((size_t)n * per_o / 8)), dim3(256), 0, st, Ro, G, row2img,
                             dst, n, per_o / 8);
        else
          hipLaunchKernelGGL(rn_mul_gate_kernel, dim3(stream_grid((size_t)n * per_o)), dim3(256), 0, st, Ro, G, row2img,
                             (const float*)nullptr, dst, n, per_o);
      };
      // S3 = R_o * (fA Q3): already written by the epilogue of the identity block walked before this one (same operand
      // format: have_s3 is only set then) — round 3 recomputed it here, one pass over R_o per stack for nothing
      float* A = s3;                                     // r1 / r5: S3, later free
      float* Bf = s3_other;                              // the other one: S1, then S0
      if (!have_s3) {
        head(b.GA.as<float>(), A);
        LRP_HIP_CHECK(hipGetLastError());
      }
      LRP_TRY(unit_backward(units[b.u3], n, row2img, A, units[b.u2].gate.as<float>(), r2.as<float>(), st, sp, !sp));              // S2
      LRP_TRY(unit_backward(units[b.u2], n, row2img, r2.as<float>(), units[b.u1].gate.as<float>(), Bf, st, sp, !sp));              // S1
      const float* taux = b.stride == 2 ? b.t_sub.as<float>() : b.t_in.as<float>();
      LRP_TRY(unit_backward(units[b.u1], n, row2img, Bf, taux, r2.as<float>(), st, sp, true));                                    // t*C1 (coarse)
      head(b.GS.as<float>(), Bf);                                                                                                  // S0
      LRP_HIP_CHECK(hipGetLastError());
      LRP_TRY(unit_backward(units[b.u0], n, row2img, Bf, taux, r3.as<float>(), st, sp, true));                                    // t*C0
      // join: R_t = t*C1 + t*C0 (scattered to the even positions behind a stride-2 block) and — one pass — the head of the
      // block walked next, S3' = R_t * GA' (an identity block: the last of the stack below)
      have_s3 = false;
      const bool fuse_next = sp && bi > 0 && blocks[bi - 1].u0 < 0 && block_split(blocks[bi - 1]) && !(b.cin & 7);
      if (sp && !(b.cin & 7)) {
        const size_t tot8 = (size_t)n * b.Hin * b.Win * b.cin / 8;
        const float* g2 = fuse_next ? blocks[bi - 1].GA.as<float>() : (const float*)nullptr;
        float* o2 = fuse_next ? A : (float*)nullptr;
        if (b.stride == 2)
          hipLaunchKernelGGL(rn_join_split_kernel<true>, dim3(stream_grid(tot8)), dim3(256), 0, st, r2.as<float>(), r3.as<float>(), cur,
                             g2, row2img, o2, n, b.Hin, b.Win, b.cin);
        else
          hipLaunchKernelGGL(rn_join_split_kernel<false>, dim3(stream_grid(tot8)), dim3(256), 0, st, r2.as<float>(), r3.as<float>(), cur,
                             g2, row2img, o2, n, b.Hin, b.Win, b.cin);
        if (fuse_next) { s3 = A; s3_other = Bf; have_s3 = true; }
      } else if (b.stride == 2) {
        const size_t tot = (size_t)n * b.Hin * b.Win * b.cin;
        hipLaunchKernelGGL(rn_scatter2_kernel, dim3(stream_grid(tot)), dim3(256), 0, st, r2.as<float>(), r3.as<float>(), cur, n,
                           b.Hin, b.Win, b.cin);
      } else {
        const size_t per_c = (size_t)b.H * b.W * b.cin;
        hipLaunchKernelGGL(rn_add_kernel, dim3(stream_grid((size_t)n * per_c)), dim3(256), 0, st, r2.as<float>(), r3.as<float>(), cur,
                           (size_t)n * per_c);
      }
      LRP_HIP_CHECK(hipGetLastError());
      Ro = cur;
      std::swap(cur, other);
    }
    // stem: pool routing * Q_stem -> T = S . W (K = stem_c, N = 294) -> 7x7/2 stencil with the x+/x- selection
    const RnUnit& s = units[0];
    const size_t tot = (size_t)n * s.Hout * s.Wout * s.cout;
    // the stem's tap GEMM (K = stem channels -> 49 taps x 6 columns) as bf16x3 like the other convs of the walk: the pool
    // routing writes its operand in split8 form (fp32 mode: plain fp32 on the fp32 MFMA)
    const bool ssplit = prec == PREC_BF16X3 && s.w_bs.p && !(s.cout & 7);
    if (ssplit)
      hipLaunchKernelGGL(rn_pool3_route_kernel<true>, dim3(stream_grid(tot / 8)), dim3(256), 0, st, Ro, pool_win.as<unsigned char>(),
                         q_stem.as<float>(), row2img, r1.as<float>(), n, s.Hout, s.Wout, s.cout);
    else
      hipLaunchKernelGGL(rn_pool3_route_kernel<false>, dim3(stream_grid(tot / 4)), dim3(256), 0, st, Ro, pool_win.as<unsigned char>(),
                         q_stem.as<float>(), row2img, r1.as<float>(), n, s.Hout, s.Wout, s.cout);
    LRP_HIP_CHECK(hipGetLastError());
    if (sw().img_fused != 0 && s.cout == 64 && (ssplit ? s.w_bs.p : s.w_b.p) && conv_npad(RN_STEM_TCOLS) >= 5 * 60 + 4 && conv_cinp(s.cout) == 64) {
      // T = S . W and the 7x7 / stride-2 shift-and-add in ONE launch, T never leaves LDS (resnet_kernels.h rn_stem_reverse_kernel);
      // LRP_IMG_FUSED=0 (and stems that are not 64 channels wide): the two-kernel form below
      const int txs = (s.Wout + RN_ST - 1) / RN_ST, tys = (s.Hout + RN_ST - 1) / RN_ST;
      if (!stem_attr_set) {
        LRP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&rn_stem_reverse_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, RN_STEM_LDS));
        LRP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&rn_stem_reverse_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, RN_STEM_LDS));
        stem_attr_set = true;
      }
      if (ssplit)
        hipLaunchKernelGGL(rn_stem_reverse_kernel<true>, dim3((unsigned)(n * txs * tys)), dim3(512), RN_STEM_LDS, st, r1.as<float>(), s.w_bs.as<float>(),
                           images.as<float>(), row2img, R_img_dev, s.Hout, s.Wout, txs, tys);
      else
        hipLaunchKernelGGL(rn_stem_reverse_kernel<false>, dim3((unsigned)(n * txs * tys)), dim3(512), RN_STEM_LDS, st, r1.as<float>(), s.w_b.as<float>(),
                           images.as<float>(), row2img, R_img_dev, s.Hout, s.Wout, txs, tys);
      LRP_HIP_CHECK(hipGetLastError());
      return LRP_OK;
    }
    ConvArgs ca{};
    ca.in = r1.as<float>(); ca.NB = n * s.Hout * s.Wout; ca.H = 1; ca.W = 1; ca.Cin = s.cout; ca.CinP = conv_cinp(s.cout);
    ca.taps = 1; ca.wpk = ssplit ? s.w_bs.as<float>() : s.w_b.as<float>(); ca.N = RN_STEM_TCOLS; ca.out = r2.as<float>();
    LRP_HIP_CHECK(conv_launch(EPI_STORE, ca, st, ssplit ? PREC_BF16X3 : PREC_FP32));
    hipLaunchKernelGGL(rn_stem_stencil_kernel, dim3(stream_grid((size_t)n * img_h * img_w)), dim3(256), 0, st, r2.as<float>(),
                       images.as<float>(), row2img, R_img_dev, n, img_h, img_w);
    LRP_HIP_CHECK(hipGetLastError());
    return LRP_OK;
  }

  // the stem's routing needs Q alone: what arrives at a0 = relu(y) through the pool is already a relevance
  // (t * C1 of the first block), so multiplying by the relu-unit gate a0*Q would count a0 twice
  DevBuf q_stem;
  DevBuf pool_win;
  bool stem_attr_set = false;

  int profile_records(int cap, double* ms_out, double* flop_out, int* n_out) {
    int k = 0;
    for (ProfileRec& p : prof) {
      float t = 0.f;
      const bool ok = hipEventSynchronize(p.e1) == hipSuccess && hipEventElapsedTime(&t, p.e0, p.e1) == hipSuccess;
      if (ok && k < cap) { ms_out[k] = t; flop_out[k] = p.flop; ++k; }
      (void)hipEventDestroy(p.e0); (void)hipEventDestroy(p.e1);
    }
    prof.clear();
    *n_out = k;
    return LRP_OK;
  }
};

}  // namespace lrp
