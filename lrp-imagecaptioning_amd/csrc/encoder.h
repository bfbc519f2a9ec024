// encoder.h — host-side orchestration of the CNN half:
//   encode():  one forward per IMAGE, caching the relevance gates G_l and Z_top
//   explain(): one reverse walk per TOKEN (batched over all tokens of the call)
// Reference semantics: LRPSequentialPresetA.analyze([X,R]) (AB:478-520) over the
// sub-model input_1 -> block5_conv3 (E:29-32); rules RR:274-322, RA:470-480.
#pragma once
#include <algorithm>
#include <cstring>
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "cnn_kernels.h"
#include "common.h"
#include "conv_igemm.h"
#include "conv_sparse.h"
#include "f16_operand.h"

namespace lrp {

struct ConvLayer {
  std::string name;
  int cin = 0, cout = 0, H = 0, W = 0;   // H,W = resolution this conv runs at
  bool pool_after = false;
  bool have_w = false, have_b = false;
  DevBuf w_fwd;    // dual-packed forward weights  (a_l | Z+_l)          [fp32 mode, and the image layer]
  DevBuf w_fwd_a;  // forward weights w, fp32                            [mixed mode: exact activation conv]
  DevBuf w_fwd_zs; // forward weights w+, split8                         [mixed mode: bf16x3 denominator conv]
  DevBuf w_fwd_as; // forward weights w, split8                                    [activation conv, LRP_PREC_BF16X3_FAST]
  DevBuf w_fwd_h;  // the dual matrix (w | w+) of w_fwd as fp16 pairs, one common scale (record wds)   [a_l and Z+_l in one pass, default]
  DevBuf wds;
  DevBuf w_bwd;    // w+ (and w- for the image layer), tap-flipped, packed for convT-as-conv
  DevBuf w_bwd_s;  // the same matrix in split8 (bf16 hi|lo) form for the bf16x3 reverse walk
  DevBuf w_bwd_full;  // full w (both signs), tap-flipped: the gradient baselines' backward-data conv (fp32)
  DevBuf w_bwd_full_s;  // the same in split8 form: backward-data conv of the fine-tune step on the bf16 matrix cores
  DevBuf w_bwd_h;     // w_bwd in fp16 split8 form [hi8 | lo8] (PREC_F16X2 reverse walk: only hi is read), and ...
  DevBuf w_bwd_frag_h;   // ... fragment-major for the weights-in-registers kernel
  DevBuf wbs;         // device record {2^k, 2^-k, norm, k} of the fp16 backward copy's power-of-two scale
  DevBuf w_fwd_il;    // the dual forward matrix with its rows interleaved per 32 channels ([w | w+] side by side): fp32 source of w_fwd_h
  std::unique_ptr<TileOrder> order{new TileOrder};   // tile-row order of this layer's reverse launch (conv_igemm.h)
  DevBuf w_bwd_frag;  // w_bwd_s fragment-major (layers whose backward conv has N = cin <= 64: weights-in-registers kernel)
  DevBuf bias;
  DevBuf G;        // [max_images][H][W][cout] relevance gate (not for the top layer)
  DevBuf P;        // [max_images][H/2][W/2][cout] pooled activations (pool_after layers; overlapped encode)
  // the gate of a pooled layer in COMPACT form (value + position per window and channel), for the consumer of the compact
  // pool interface; written by pool_gate_split_kernel, valid for the encode whose number gc_epoch holds
  DevBuf Gc, Gpos;
  long gc_epoch = -1;
  // the full-resolution gate G of a pooled layer is current for the encode whose number this holds: the fused pool epilogue
  // writes the compact form only, Encoder::full_gate() expands it for the walks that read G (cnn_kernels.h pool_gate_expand_kernel)
  long gfull_epoch = -1;
  // 2:4-sparse consumer of the pooled boundary behind this layer (conv_sparse.h; layers with cin % 256 == 0 whose output is pooled:
  // VGG16 block3_conv3, block4_conv3): the four class arrangements of w+ and, per encode, the index planes of the pool's positions
  DevBuf w_sp, idxp;
  long idx_epoch = -1;
  bool sparse_ok() const { return pool_after && conv_sparse_supports(cin, cout, H / 2, W / 2) && !(H & 1) && !(W & 1); }
  std::vector<float> raw_w, raw_b;   // host copies as set (HWIO / (cout,)): the fine-tune step's master weights start here
  DevBuf raw_w_dev, raw_b_dev;       // the same when the weights arrived through lrp_set_weight_dev (no host copy exists then)
  DevBuf fnorm;    // {largest absolute row sum of w, max|b|}: bound behind the scale of the pairs this layer emits (fwd_scale_kernel)
  bool norm_dirty = true;
  DevBuf Akeep;    // fine-tune step only: a_l of the layers whose output is the next conv's input (no pool after); the
                   // LRP path turns that storage into the gate in place
  size_t act_elems() const { return (size_t)H * W * cout; }
};

struct ProfileRec {
  hipEvent_t e0, e1;
  double flop;
};

struct Encoder {
  int img_h = 0, img_w = 0, max_images = 0, max_tokens = 0;
  int top_h = 0, top_w = 0, top_c = 0;
  std::vector<ConvLayer> layers;
  DevBuf images;           // [max_images][H][W][3]   (x of the image layer, needed by img_stencil_kernel)
  DevBuf a1;               // im2col of the image layer [max_images*H*W][64]
  DevBuf bufX, bufA, bufZ; // forward ping-pong (per call, all images)
  DevBuf bufXs;            // split8 copy of the current conv input (mixed-precision forward)
  DevBuf bufXl;            // its [h | l] companion (three-way split forward product)
  DevBuf feat;             // [max_images][top_h*top_w][top_c]  top activations (== CNN features)
  DevBuf ztop;             // [max_images][top...] Z+ of the top layer
  DevBuf s0, s1;           // reverse-walk ping-pong [max_tokens][biggest layer]
  int encoded = 0;         // images currently cached
  bool features_only = false;
  bool profile = false;
  int prec = PREC_BF16X3;  // arithmetic of the per-token reverse walk (lrp_set_precision); falls back to fp32 for widths % 8 != 0
  const int* row2img_host = nullptr;   // host copy of the NEXT explain call's token -> image map (one-shot; lets the launcher
                                       // order the tiles so that an image's gates are fetched once, conv_igemm.h TileOrder)
  bool walk_f16 = false;   // LRP_PREC_F16X2 (opt-in): the LRP reverse walk on fp16 pairs, 2 MFMAs per product below the top block.
  // Which layers take the two-MFMA form in that mode (reverse launch through layer li AND its forward denominators Z+_li: the
  // two always go together).  -1 = the built-in rule (every layer up to the last pool whose sums have >= 576 products);
  // otherwise bit li (lrp_set_fast_layers: a per-model choice, e.g. from calibration.py's measured per-layer error).
  int64_t t2_mask_user = -1;
  bool two_term(int li) const {
    if (li <= 0 || li >= (int)layers.size()) return false;
    if (t2_mask_user >= 0) return ((t2_mask_user >> li) & 1) != 0;
    int last_pool = -1;
    for (size_t q = 0; q < layers.size(); ++q)
      if (layers[q].pool_after) last_pool = (int)q;
    return li <= last_pool && 9 * layers[li].cout >= 576 && 9 * layers[li].cin >= 576;   // (narrow test nets: too few products to average over)
  }
                           // Its parity depends on the weight statistics (one fp16 per weight: worst case 2^-12 per product, above the
                           // 1e-4 bar; tests/test_gpu_stress_parity.py), so the default is the three-MFMA split-bf16 walk.
  DevBuf sp_scp;                      // sparse consumers: S_c of the call as chunk-major pairs (conv_sparse.h)
  DevBuf act_max, act_unscale;        // fp16-pair forward: per layer ACT_MAX_SLOTS maxima of its output / 2^-k of its input
  DevBuf out_scale;                   // ... and 2^k of the pairs a layer emits for its consumer (no split pass in between)
  static bool fwd_emit() { return sw().fwd_emit != 0; }   // LRP_FWD_EMIT=0: split passes between the convs as in round 2
  DevBuf tok_exp, tok_max, tok_fac;   // its per-token scale exponents / measured maxima [layers + 1][max_tokens], factors [max_tokens]
  std::vector<ProfileRec> prof;
  // Overlapped encode (mixed-precision mode): the caller's stream runs only the activation chain a_1..a_top (what
  // the decoder needs); the denominators Z+_l and the gates G_l — needed by explain() only — run on `side` behind
  // it, i.e. concurrently with the latency-bound decoder replay the caller enqueues next.
  hipStream_t side = nullptr;
  hipEvent_t ev_fwd = nullptr, ev_gates = nullptr;
  bool gates_pending = false;

  ~Encoder() {
    if (side) { (void)hipStreamSynchronize(side); (void)hipStreamDestroy(side); }
    if (ev_fwd) (void)hipEventDestroy(ev_fwd);
    if (ev_gates) (void)hipEventDestroy(ev_gates);
  }
  static bool img_fused() { return sw().img_fused != 0; }   // LRP_IMG_FUSED=0: separate T GEMM + img_stencil_kernel
  // First layer whose ACTIVATION conv runs split-bf16 too.  A ~1e-5 relative error in a_l is harmless by itself, but
  // upstream of a 2x2 max-pool it flips the arg-max of near-tied windows (~1e-5 of them), and a flipped window moves
  // its whole relevance to a neighbour pixel: measured on VGG16, relative L1 of the heat-maps 5e-6 ... 3.6e-5 instead
  // of 5.6e-6.  Default: none (splitting only the layers behind the last pool is flip-free but makes the features the
  // decoder consumes 10x less exact, 7.4e-7 -> 7.8e-6, for 0.5 ms); lrp_set_precision(LRP_PREC_BF16X3_FAST): every layer
  // but the image layer (-5 ms, heat-map parity <= 4e-5).
  bool fwd_fast = false;
  int fwd_split_from() const { return fwd_fast ? 1 : 1000; }
  // Activation chain and denominators in ONE pass on the fp16 MFMA: operands as fp16 pairs hi + lo (22 mantissa bits, x scaled by
  // a power of two per layer and image), product hi*hi' + hi*lo' + lo*hi' in three MFMAs with blocked fp32 accumulation,
  // weights (w | w+) stacked along N.  (Rounds 1-2 kept the earlier arrangements — two-pass three-way bf16 split, single
  // activation conv + side-stream Z+ chain — behind LRP_FWD_F16 / LRP_FWD_DUAL / LRP_FWD_X6; removed in round 4: untested.)
  // fp16-pair dual forward with interleaved weight rows: a_l and the gate G_l = a_l / safe(Z+_l) leave the conv's epilogue
  // together where no pool follows (conv_igemm.h ConvArgs::dual_il) — no Z+ tensor, no gate pass.  LRP_FWD_IL=0: stacked rows.
  static bool fwd_il() { return sw().fwd_il != 0; }
  // the image layer's fp32 dual GEMM with interleaved rows: gate, pairs and maximum leave its epilogue (no gate / absmax / split pass)
  static bool image_layer_interleaved(const ConvLayer& L) { return fwd_il() && !(L.cout & 31) && conv_npad(2 * L.cout) == 2 * L.cout && !L.pool_after; }
  static bool dual_interleaved(const ConvLayer& L) { return fwd_il() && !(L.cout & 31) && conv_npad(2 * L.cout) == 2 * L.cout; }
  int init(const lrp_config& c, int64_t* total) {
    img_h = c.img_h; img_w = c.img_w; max_images = c.max_images; max_tokens = c.max_tokens;
    if (c.n_conv < 1 || c.n_conv > LRP_MAX_CONV) return fail(LRP_ERR_INVALID, "n_conv=%d out of range", c.n_conv);
    if (c.conv_cin[0] != 3) return fail(LRP_ERR_UNSUPPORTED, "first conv must read a 3-channel image");
    int H = img_h, W = img_w;
    size_t max_act = 0;
    layers.resize(c.n_conv);
    for (int i = 0; i < c.n_conv; ++i) {
      ConvLayer& L = layers[i];
      L.name = c.conv_name[i];
      L.cin = c.conv_cin[i]; L.cout = c.conv_cout[i]; L.H = H; L.W = W;
      L.pool_after = c.conv_pool_after[i] != 0;
      if (i > 0 && L.cin != layers[i - 1].cout) return fail(LRP_ERR_INVALID, "conv %d: cin != previous cout", i);
      if (L.cout % 4 != 0) return fail(LRP_ERR_UNSUPPORTED, "conv %d: cout must be a multiple of 4", i);
      if (i == c.n_conv - 1 && L.pool_after) return fail(LRP_ERR_UNSUPPORTED, "encoder must end with a conv layer");
      if (L.act_elems() > max_act) max_act = L.act_elems();
      if (L.pool_after) {
        if ((H & 1) || (W & 1)) return fail(LRP_ERR_UNSUPPORTED, "odd resolution before a 2x2 pool");
        H >>= 1; W >>= 1;
      }
    }
    const ConvLayer& T = layers.back();
    top_h = T.H; top_w = T.W; top_c = T.cout;
    if (top_h * top_w != c.L || top_c != c.D)
      return fail(LRP_ERR_INVALID, "encoder output (%d x %d x %d) does not match L=%d, D=%d", top_h, top_w, top_c, c.L, c.D);
    const size_t B = (size_t)max_images, NT = (size_t)max_tokens;
    const size_t max_tok_act = std::max(max_act, (size_t)img_h * img_w * IMG_T_COLS);   // T of the image layer
    LRP_TRY(images.alloc(B * img_h * img_w * 3 * sizeof(float), total));
    LRP_TRY(a1.alloc(B * img_h * img_w * 64 * sizeof(float), total));
    LRP_TRY(bufX.alloc(B * max_act * sizeof(float), total));
    LRP_TRY(bufA.alloc(B * max_act * sizeof(float), total));
    LRP_TRY(bufZ.alloc(B * max_act * sizeof(float), total));
    LRP_TRY(bufXs.alloc(B * max_act * sizeof(float), total));
    LRP_TRY(bufXl.alloc(B * max_act * sizeof(float), total));
    LRP_TRY(feat.alloc(B * T.act_elems() * sizeof(float), total));
    LRP_TRY(ztop.alloc(B * T.act_elems() * sizeof(float), total));
    LRP_TRY(s0.alloc(NT * max_tok_act * sizeof(float), total));
    LRP_TRY(s1.alloc(NT * max_tok_act * sizeof(float), total));
    for (size_t i = 0; i + 1 < layers.size(); ++i) LRP_TRY(layers[i].G.alloc(B * layers[i].act_elems() * sizeof(float), total));
    for (size_t i = 0; i + 1 < layers.size(); ++i) {
      ConvLayer& Lc = layers[i];                         // (the condition of Encoder::explain for the compact interface, weights aside)
      if (Lc.pool_after && !(Lc.cout & 7) && !(Lc.H & 1) && !(Lc.W & 1) &&
          ((Lc.cin <= 64 && conv_cinp(Lc.cout) <= 64) || conv_pick_tile(Lc.cin).BN == 128)) {
        LRP_TRY(Lc.Gc.alloc(B * Lc.act_elems() / 4 * sizeof(float), total));
        LRP_TRY(Lc.Gpos.alloc(B * Lc.act_elems() / 4, total));
      }
    }
    for (size_t i = 0; i + 1 < layers.size(); ++i)
      if (layers[i].pool_after) LRP_TRY(layers[i].P.alloc(B * layers[i].act_elems() / 4 * sizeof(float), total));
    {  // sparse consumers: index planes per image, one chunk-major copy of S_c per call (the largest of them)
      size_t mx = 0;
      for (size_t i = 1; i + 1 < layers.size(); ++i) {
        ConvLayer& Lc = layers[i];
        if (!Lc.sparse_ok() || !Lc.Gpos.p || layers[i - 1].pool_after) continue;
        LRP_TRY(Lc.idxp.alloc(conv_sparse_index_words((int)B, Lc.H / 2, Lc.W / 2, Lc.cout) * sizeof(unsigned), total));
        mx = std::max(mx, NT * Lc.act_elems() / 4);
      }
      if (mx) LRP_TRY(sp_scp.alloc(mx * sizeof(float), total));
    }
    {
      // lowest priority: the side work is throughput work that should only fill what the caller's stream leaves idle
      int lo = 0, hi = 0;
      LRP_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
      LRP_HIP_CHECK(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, lo));
    }
    LRP_HIP_CHECK(hipEventCreateWithFlags(&ev_fwd, hipEventDisableTiming));
    LRP_HIP_CHECK(hipEventCreateWithFlags(&ev_gates, hipEventDisableTiming));
    // scale records of the fp16-pair forward, per layer AND image (the emitting forward scales every image by its own
    // maxima; +1 level: the images themselves)
    LRP_TRY(act_max.alloc((layers.size() + 1) * B * ACT_MAX_SLOTS * sizeof(unsigned), total));
    LRP_TRY(act_unscale.alloc((layers.size() + 1) * B * sizeof(float), total));
    LRP_TRY(out_scale.alloc((layers.size() + 1) * B * sizeof(float), total));
    for (ConvLayer& L : layers) LRP_TRY(L.fnorm.alloc(2 * sizeof(float), total));
    return LRP_OK;
  }

  // Fine-tune step: the weight gradient of layer l needs a_{l-1}; keep what the LRP caches drop.
  bool keep_acts = false;
  int enable_keep_acts(int64_t* total) {
    if (keep_acts) return LRP_OK;
    for (size_t li = 0; li + 1 < layers.size(); ++li) {
      ConvLayer& L = layers[li];
      if (!L.pool_after) LRP_TRY(L.Akeep.alloc((size_t)max_images * L.act_elems() * sizeof(float), total));
    }
    keep_acts = true;
    return LRP_OK;
  }
  // input of conv li as the last encode left it (li >= 1; the image itself for li = 0)
  const float* layer_input(int li) const {
    if (li == 0) return images.as<float>();
    const ConvLayer& P = layers[li - 1];
    return P.pool_after ? P.P.as<float>() : P.Akeep.as<float>();
  }

  int find_layer(const std::string& nm) const {
    for (size_t i = 0; i < layers.size(); ++i)
      if (layers[i].name == nm) return (int)i;
    return -1;
  }

  // "<name>_W": HWIO (3,3,cin,cout) -> split by sign (RR:256-260), pack, upload.
  int set_conv_weight(int li, const float* w, int64_t* total) {
    ConvLayer& L = layers[li];
    const size_t nW = (size_t)9 * L.cin * L.cout;
    if (w != L.raw_w.data()) L.raw_w.assign(w, w + nW);
    L.raw_w_dev.release();
    std::vector<float> wp(nW), wn(nW);
    for (size_t i = 0; i < nW; ++i) { wp[i] = w[i] >= 0.f ? w[i] : 0.f; wn[i] = w[i] < 0.f ? w[i] : 0.f; }
    std::vector<float> pk;
    if (li == 0) {
      // forward: 1-tap GEMM over the im2col matrix A1[m][64]; rows = [a_1 (cout) | Z_1 (cout)]
      const int Np = conv_npad(2 * L.cout), K = 64;
      pk.assign((size_t)Np * K, 0.f);
      for (int k = 0; k < 27; ++k)
        for (int co = 0; co < L.cout; ++co) {
          const float v = w[(size_t)k * L.cout + co];
          pk[(size_t)co * K + k] = v;                       // a_1 = (x+ + x-) . w
          pk[(size_t)co * K + 32 + k] = v;
          pk[(size_t)(L.cout + co) * K + k] = wp[(size_t)k * L.cout + co];        // Z_1 = x+.w+ + x-.w-
          pk[(size_t)(L.cout + co) * K + 32 + k] = wn[(size_t)k * L.cout + co];
        }
      LRP_TRY(L.w_fwd.alloc(pk.size() * sizeof(float), total));
      LRP_HIP_CHECK(hipMemcpy(L.w_fwd.p, pk.data(), pk.size() * sizeof(float), hipMemcpyHostToDevice));
      if (!image_layer_interleaved(L)) L.w_fwd_il.release();   // (the layout of a layer's copies is decided HERE; encode() asks which copies exist)
      if (image_layer_interleaved(L)) {
        LRP_TRY(L.w_fwd_il.alloc(pk.size() * sizeof(float), total));
        hipLaunchKernelGGL(dual_interleave_rows_kernel, dim3(stream_grid((size_t)2 * L.cout * K)), dim3(256), 0, nullptr, L.w_fwd.as<float>(),
                           L.w_fwd_il.as<float>(), L.cout, K);
        LRP_HIP_CHECK(hipGetLastError());
        LRP_HIP_CHECK(hipStreamSynchronize(nullptr));
        LRP_TRY(L.w_fwd_h.alloc(pk.size() * sizeof(float), total));
        LRP_TRY(make_f16_operand(L.w_fwd_il.as<float>(), pk.size(), 0, 0, L.w_fwd_h, L.wds, total, nullptr));
      }
      // backward at the image: tap-expanded channel reduction, 54 = 9 taps x (3 with w+ | 3 with w-)
      // columns, K = cout; the 3x3 shift-and-add happens in img_stencil_kernel.
      const int Npb = conv_npad(IMG_T_COLS), Kb = conv_cinp(L.cout);
      pk.assign((size_t)Npb * Kb, 0.f);
      for (int t = 0; t < 9; ++t)
        for (int c = 0; c < 3; ++c)
          for (int co = 0; co < L.cout; ++co) {
            pk[(size_t)(t * 6 + c) * Kb + co] = wp[((size_t)t * 3 + c) * L.cout + co];
            pk[(size_t)(t * 6 + 3 + c) * Kb + co] = wn[((size_t)t * 3 + c) * L.cout + co];
          }
      LRP_TRY(L.w_bwd.alloc(pk.size() * sizeof(float), total));
      LRP_HIP_CHECK(hipMemcpy(L.w_bwd.p, pk.data(), pk.size() * sizeof(float), hipMemcpyHostToDevice));
      {
        std::vector<float> sp(pk.size());
        pack_split8(pk.data(), pk.size(), sp.data());
        LRP_TRY(L.w_bwd_s.alloc(sp.size() * sizeof(float), total));
        LRP_HIP_CHECK(hipMemcpy(L.w_bwd_s.p, sp.data(), sp.size() * sizeof(float), hipMemcpyHostToDevice));
        LRP_TRY(L.w_bwd_h.alloc(sp.size() * sizeof(float), total));
        LRP_TRY(make_f16_operand(L.w_bwd.as<float>(), sp.size(), 0, 0, L.w_bwd_h, L.wbs, total, nullptr));
      }
      // gradient baselines: the same tap expansion with the whole w in the "+" columns
      pk.assign((size_t)Npb * Kb, 0.f);
      for (int t = 0; t < 9; ++t)
        for (int c = 0; c < 3; ++c)
          for (int co = 0; co < L.cout; ++co) pk[(size_t)(t * 6 + c) * Kb + co] = w[((size_t)t * 3 + c) * L.cout + co];
      LRP_TRY(L.w_bwd_full.alloc(pk.size() * sizeof(float), total));
      LRP_HIP_CHECK(hipMemcpy(L.w_bwd_full.p, pk.data(), pk.size() * sizeof(float), hipMemcpyHostToDevice));
    } else {
      const int Np = conv_npad(2 * L.cout), K = 9 * conv_cinp(L.cin);
      pk.assign((size_t)Np * K, 0.f);
      pack_conv_fwd(w, 9, L.cin, L.cout, 0, Np, pk.data());
      pack_conv_fwd(wp.data(), 9, L.cin, L.cout, L.cout, Np, pk.data());    // input >= 0: Z = x.w+ + b
      LRP_TRY(L.w_fwd.alloc(pk.size() * sizeof(float), total));
      LRP_HIP_CHECK(hipMemcpy(L.w_fwd.p, pk.data(), pk.size() * sizeof(float), hipMemcpyHostToDevice));
      LRP_TRY(L.w_fwd_h.alloc(pk.size() * sizeof(float), total));
      if (dual_interleaved(L)) {
        std::vector<float> il(pk.size(), 0.f);            // rows in blocks of 32: [w | w+] of the same 32 channels
        for (int c = 0; c < L.cout; ++c)
          for (int half = 0; half < 2; ++half)
            memcpy(&il[(size_t)(64 * (c / 32) + 32 * half + (c & 31)) * K], &pk[(size_t)(half * L.cout + c) * K], (size_t)K * sizeof(float));
        LRP_TRY(L.w_fwd_il.alloc(il.size() * sizeof(float), total));
        LRP_HIP_CHECK(hipMemcpy(L.w_fwd_il.p, il.data(), il.size() * sizeof(float), hipMemcpyHostToDevice));
      } else {
        L.w_fwd_il.release();
      }
      LRP_TRY(make_f16_operand(L.w_fwd_il.p ? L.w_fwd_il.as<float>() : L.w_fwd.as<float>(), pk.size(), 0, 0, L.w_fwd_h, L.wds,
                               total, nullptr));
      {  // mixed-precision forward: w (fp32) and w+ (split8) as separate N = cout matrices
        const int Npa = conv_npad(L.cout);
        std::vector<float> pa((size_t)Npa * K, 0.f), pz((size_t)Npa * K, 0.f), pzs((size_t)Npa * K);
        pack_conv_fwd(w, 9, L.cin, L.cout, 0, Npa, pa.data());
        pack_conv_fwd(wp.data(), 9, L.cin, L.cout, 0, Npa, pz.data());
        pack_split8(pz.data(), pz.size(), pzs.data());
        LRP_TRY(L.w_fwd_a.alloc(pa.size() * sizeof(float), total));
        LRP_HIP_CHECK(hipMemcpy(L.w_fwd_a.p, pa.data(), pa.size() * sizeof(float), hipMemcpyHostToDevice));
        LRP_TRY(L.w_fwd_zs.alloc(pzs.size() * sizeof(float), total));
        LRP_HIP_CHECK(hipMemcpy(L.w_fwd_zs.p, pzs.data(), pzs.size() * sizeof(float), hipMemcpyHostToDevice));
        pack_split8(pa.data(), pa.size(), pzs.data());
        LRP_TRY(L.w_fwd_as.alloc(pzs.size() * sizeof(float), total));
        LRP_HIP_CHECK(hipMemcpy(L.w_fwd_as.p, pzs.data(), pzs.size() * sizeof(float), hipMemcpyHostToDevice));
      }
      const int Npb = conv_npad(L.cin), Kb = 9 * conv_cinp(L.cout);
      pk.assign((size_t)Npb * Kb, 0.f);
      pack_conv_bwd(wp.data(), 9, L.cin, L.cout, 0, pk.data());
      LRP_TRY(L.w_bwd.alloc(pk.size() * sizeof(float), total));
      LRP_HIP_CHECK(hipMemcpy(L.w_bwd.p, pk.data(), pk.size() * sizeof(float), hipMemcpyHostToDevice));
      {
        std::vector<float> sp(pk.size());
        pack_split8(pk.data(), pk.size(), sp.data());
        LRP_TRY(L.w_bwd_s.alloc(sp.size() * sizeof(float), total));
        LRP_HIP_CHECK(hipMemcpy(L.w_bwd_s.p, sp.data(), sp.size() * sizeof(float), hipMemcpyHostToDevice));
        if (L.sparse_ok() && L.idxp.p) {
          LRP_TRY(L.w_sp.alloc(conv_sparse_weight_floats(L.cin, L.cout) * sizeof(float), total));
          LRP_HIP_CHECK(conv_sparse_pack(L.w_bwd.as<float>(), L.w_sp.as<float>(), L.cin, L.cout, nullptr));
        }
        if (Npb == 64) {
          std::vector<float> fr((size_t)64 * Kb);
          pack_frag64(sp.data(), 9, conv_cinp(L.cout), fr.data());
          LRP_TRY(L.w_bwd_frag.alloc(fr.size() * sizeof(float), total));
          LRP_HIP_CHECK(hipMemcpy(L.w_bwd_frag.p, fr.data(), fr.size() * sizeof(float), hipMemcpyHostToDevice));
        }
        // fp16 copies (PREC_F16X2), their scale and norm: derived on the device from the fp32 matrix just uploaded
        LRP_TRY(L.w_bwd_h.alloc(sp.size() * sizeof(float), total));
        if (Npb == 64) LRP_TRY(L.w_bwd_frag_h.alloc((size_t)64 * Kb * sizeof(float), total));
        LRP_TRY(make_f16_operand(L.w_bwd.as<float>(), sp.size(), Npb, Kb, L.w_bwd_h, L.wbs, total, nullptr));
        if (Npb == 64) {
          hipLaunchKernelGGL(pack_frag64_dev_kernel, dim3(stream_grid((size_t)conv_cinp(L.cout) / 32 * 9 * 512)), dim3(256), 0, nullptr,
                             L.w_bwd_h.as<float>(), L.w_bwd_frag_h.as<float>(), conv_cinp(L.cout));
          LRP_HIP_CHECK(hipGetLastError());
          LRP_HIP_CHECK(hipStreamSynchronize(nullptr));
        }
      }
      pk.assign((size_t)Npb * Kb, 0.f);
      pack_conv_bwd(w, 9, L.cin, L.cout, 0, pk.data());
      LRP_TRY(L.w_bwd_full.alloc(pk.size() * sizeof(float), total));
      LRP_HIP_CHECK(hipMemcpy(L.w_bwd_full.p, pk.data(), pk.size() * sizeof(float), hipMemcpyHostToDevice));
      {
        std::vector<float> sp(pk.size());
        pack_split8(pk.data(), pk.size(), sp.data());
        LRP_TRY(L.w_bwd_full_s.alloc(sp.size() * sizeof(float), total));
        LRP_HIP_CHECK(hipMemcpy(L.w_bwd_full_s.p, sp.data(), sp.size() * sizeof(float), hipMemcpyHostToDevice));
      }
    }
    L.have_w = true;
    L.norm_dirty = true;
    return LRP_OK;
  }

  // ---- operand copies built ON THE DEVICE from device weights: lrp_set_weight_dev (the multi-GPU start-up path: the
  // bundle arrives over RCCL/xGMI and never visits the host) and the fine-tune step (weights change every iteration).
  DevBuf pack_tmp;                                     // scratch of the device packers (largest forward matrix)
  DevBuf f16_slots;                                    // ACT_MAX_SLOTS maxima while a weight matrix' fp16 copy is made
  int make_f16_operand(const float* src, size_t n_floats, int rows, int K, DevBuf& dst, DevBuf& wsc, int64_t* total, hipStream_t st,
                       bool sync = true) {
    return lrp::make_f16_operand(f16_slots, src, n_floats, rows, K, dst, wsc, total, st, sync);      // f16_operand.h
  }
  int alloc_conv_operands(int li, int64_t* total, hipStream_t st) {
    ConvLayer& L = layers[li];
    auto mk = [&](DevBuf& d, size_t floats) -> int {
      if (d.p && d.bytes == floats * sizeof(float)) return LRP_OK;
      LRP_TRY(d.alloc(floats * sizeof(float), total));
      LRP_HIP_CHECK(hipMemsetAsync(d.p, 0, d.bytes, st));          // padding rows / columns stay zero
      return LRP_OK;
    };
    if (li == 0) {
      const size_t nb = (size_t)conv_npad(IMG_T_COLS) * conv_cinp(L.cout);
      LRP_TRY(mk(L.w_fwd, (size_t)conv_npad(2 * L.cout) * 64));
      if (image_layer_interleaved(L)) {
        LRP_TRY(mk(L.w_fwd_il, (size_t)conv_npad(2 * L.cout) * 64));
        LRP_TRY(mk(L.w_fwd_h, (size_t)conv_npad(2 * L.cout) * 64));
      } else {
        L.w_fwd_il.release();
      }
      LRP_TRY(mk(L.w_bwd, nb)); LRP_TRY(mk(L.w_bwd_s, nb)); LRP_TRY(mk(L.w_bwd_full, nb)); LRP_TRY(mk(L.w_bwd_h, nb));
      return LRP_OK;
    }
    const size_t Kf = (size_t)9 * conv_cinp(L.cin), Kb = (size_t)9 * conv_cinp(L.cout);
    const size_t nf = (size_t)conv_npad(L.cout) * Kf, nb = (size_t)conv_npad(L.cin) * Kb;
    LRP_TRY(mk(L.w_fwd, (size_t)conv_npad(2 * L.cout) * Kf)); LRP_TRY(mk(L.w_fwd_h, (size_t)conv_npad(2 * L.cout) * Kf));
    if (dual_interleaved(L)) LRP_TRY(mk(L.w_fwd_il, (size_t)conv_npad(2 * L.cout) * Kf));
    else L.w_fwd_il.release();
    LRP_TRY(mk(L.w_fwd_a, nf)); LRP_TRY(mk(L.w_fwd_zs, nf)); LRP_TRY(mk(L.w_fwd_as, nf));
    LRP_TRY(mk(L.w_bwd, nb)); LRP_TRY(mk(L.w_bwd_s, nb)); LRP_TRY(mk(L.w_bwd_full, nb)); LRP_TRY(mk(L.w_bwd_full_s, nb));
    LRP_TRY(mk(L.w_bwd_h, nb));
    if (L.sparse_ok() && L.idxp.p) LRP_TRY(mk(L.w_sp, conv_sparse_weight_floats(L.cin, L.cout)));
    if (conv_npad(L.cin) == 64) { LRP_TRY(mk(L.w_bwd_frag, (size_t)64 * Kb)); LRP_TRY(mk(L.w_bwd_frag_h, (size_t)64 * Kb)); }
    if (pack_tmp.bytes < nf * sizeof(float)) LRP_TRY(pack_tmp.alloc(nf * sizeof(float), total));
    return LRP_OK;
  }
  // every operand copy of layer li from w_dev (HWIO, device); the buffers exist (set_conv_weight or alloc_conv_operands)
  int repack_conv_weight_from_device(int li, const float* w_dev, float* tmp, hipStream_t st) {
    ConvLayer& L = layers[li];
    if (li == 0) {
      const int Npb = conv_npad(IMG_T_COLS), Kb = conv_cinp(L.cout);
      hipLaunchKernelGGL(pack_image_layer_dev_kernel, dim3((27 * L.cout + 255) / 256), dim3(256), 0, st, w_dev, L.w_fwd.as<float>(),
                         L.w_bwd.as<float>(), L.w_bwd_full.as<float>(), L.cout, Kb);
      const size_t nb = (size_t)Npb * Kb;
      if (L.w_fwd_il.p) {
        hipLaunchKernelGGL(dual_interleave_rows_kernel, dim3(stream_grid((size_t)2 * L.cout * 64)), dim3(256), 0, st, L.w_fwd.as<float>(),
                           L.w_fwd_il.as<float>(), L.cout, 64);
        LRP_TRY(make_f16_operand(L.w_fwd_il.as<float>(), (size_t)conv_npad(2 * L.cout) * 64, 0, 0, L.w_fwd_h, L.wds, nullptr, st, false));
      }
      hipLaunchKernelGGL(split_copy_kernel, dim3(stream_grid(nb / 8)), dim3(256), 0, st, L.w_bwd.as<float>(), L.w_bwd_s.as<float>(), nb / 8);
      LRP_TRY(make_f16_operand(L.w_bwd.as<float>(), nb, 0, 0, L.w_bwd_h, L.wbs, nullptr, st, false));
      LRP_HIP_CHECK(hipGetLastError());
      return LRP_OK;
    }
    const int CPi = conv_cinp(L.cin), CPo = conv_cinp(L.cout);
    const int Np2 = conv_npad(2 * L.cout), Npa = conv_npad(L.cout), Npb = conv_npad(L.cin);
    auto pack = [&](float* dst, int bwd, int rows, int dual, int pos) {
      const size_t tot = (size_t)rows * 9 * (bwd ? CPo : CPi);
      hipLaunchKernelGGL(pack_conv_dev_kernel, dim3(stream_grid(tot)), dim3(256), 0, st, w_dev, dst, bwd, L.cin, L.cout,
                         bwd ? CPo : CPi, rows, dual, pos);
    };
    auto split = [&](const float* src, float* dst, size_t n) {
      hipLaunchKernelGGL(split_copy_kernel, dim3(stream_grid(n / 8)), dim3(256), 0, st, src, dst, n / 8);
    };
    const size_t nf = (size_t)Npa * 9 * CPi, nb = (size_t)Npb * 9 * CPo;
    pack(L.w_fwd.as<float>(), 0, Np2, 1, 0);
    if (L.w_fwd_il.p) pack(L.w_fwd_il.as<float>(), 0, Np2, 2, 0);
    LRP_TRY(make_f16_operand(L.w_fwd_il.p ? L.w_fwd_il.as<float>() : L.w_fwd.as<float>(), (size_t)Np2 * 9 * CPi, 0, 0, L.w_fwd_h,
                             L.wds, nullptr, st, false));
    pack(L.w_fwd_a.as<float>(), 0, Npa, 0, 0);
    pack(tmp, 0, Npa, 0, 1);
    split(tmp, L.w_fwd_zs.as<float>(), nf);
    split(L.w_fwd_a.as<float>(), L.w_fwd_as.as<float>(), nf);
    pack(L.w_bwd.as<float>(), 1, Npb, 0, 1);
    split(L.w_bwd.as<float>(), L.w_bwd_s.as<float>(), nb);
    if (L.w_sp.p) LRP_HIP_CHECK(conv_sparse_pack(L.w_bwd.as<float>(), L.w_sp.as<float>(), L.cin, L.cout, st));
    if (L.w_bwd_frag.p)
      hipLaunchKernelGGL(pack_frag64_dev_kernel, dim3(stream_grid((size_t)CPo / 32 * 9 * 512)), dim3(256), 0, st, L.w_bwd_s.as<float>(),
                         L.w_bwd_frag.as<float>(), CPo);
    LRP_TRY(make_f16_operand(L.w_bwd.as<float>(), nb, Npb, 9 * CPo, L.w_bwd_h, L.wbs, nullptr, st, false));
    if (L.w_bwd_frag_h.p)
      hipLaunchKernelGGL(pack_frag64_dev_kernel, dim3(stream_grid((size_t)CPo / 32 * 9 * 512)), dim3(256), 0, st, L.w_bwd_h.as<float>(),
                         L.w_bwd_frag_h.as<float>(), CPo);
    pack(L.w_bwd_full.as<float>(), 1, Npb, 0, 0);
    split(L.w_bwd_full.as<float>(), L.w_bwd_full_s.as<float>(), nb);
    LRP_HIP_CHECK(hipGetLastError());
    return LRP_OK;
  }
  // fine-tune step: layer li (weights and bias) from the trainer's master buffer
  int repack_conv_from_device(int li, const float* w_dev, const float* b_dev, float* tmp, hipStream_t st) {
    ConvLayer& L = layers[li];
    if (!L.have_w || !L.have_b) return fail(LRP_ERR_STATE, "layer %d has no operand copies to rebuild", li);
    LRP_TRY(repack_conv_weight_from_device(li, w_dev, tmp, st));
    L.norm_dirty = true;
    LRP_HIP_CHECK(hipMemcpyAsync(L.bias.p, b_dev, (size_t)L.cout * sizeof(float), hipMemcpyDeviceToDevice, st));
    L.raw_w.clear(); L.raw_b.clear();               // stale from here on (the trainer's master buffer is the truth)
    L.raw_w_dev.release(); L.raw_b_dev.release();
    return LRP_OK;
  }
  // lrp_set_weight_dev: "<name>_W" / "<name>_b" from device memory — D2D copy, then the device packers; no host round trip
  int set_conv_weight_dev(int li, const float* w_dev, int64_t* total, hipStream_t st) {
    ConvLayer& L = layers[li];
    const size_t nW = (size_t)9 * L.cin * L.cout;
    if (gates_pending) {                               // the side stream may still read the operand copies we replace
      LRP_HIP_CHECK(hipStreamWaitEvent(st, ev_gates, 0));
      gates_pending = false;
    }
    if (!L.raw_w_dev.p) LRP_TRY(L.raw_w_dev.alloc(nW * sizeof(float), total));
    LRP_HIP_CHECK(hipMemcpyAsync(L.raw_w_dev.p, w_dev, nW * sizeof(float), hipMemcpyDeviceToDevice, st));
    L.raw_w.clear();
    LRP_TRY(alloc_conv_operands(li, total, st));
    LRP_TRY(repack_conv_weight_from_device(li, L.raw_w_dev.as<float>(), pack_tmp.as<float>(), st));
    L.have_w = true;
    L.norm_dirty = true;
    encoded = 0;                                       // caches belong to the old weights
    return LRP_OK;
  }
  int set_conv_bias_dev(int li, const float* b_dev, int64_t* total, hipStream_t st) {
    ConvLayer& L = layers[li];
    if (!L.raw_b_dev.p) LRP_TRY(L.raw_b_dev.alloc((size_t)L.cout * sizeof(float), total));
    if (!L.bias.p) LRP_TRY(L.bias.alloc((size_t)L.cout * sizeof(float), total));
    LRP_HIP_CHECK(hipMemcpyAsync(L.raw_b_dev.p, b_dev, (size_t)L.cout * sizeof(float), hipMemcpyDeviceToDevice, st));
    LRP_HIP_CHECK(hipMemcpyAsync(L.bias.p, b_dev, (size_t)L.cout * sizeof(float), hipMemcpyDeviceToDevice, st));
    L.raw_b.clear();
    L.have_b = true;
    L.norm_dirty = true;
    encoded = 0;
    return LRP_OK;
  }

  int set_conv_bias(int li, const float* b, int64_t* total) {
    ConvLayer& L = layers[li];
    if (b != L.raw_b.data()) L.raw_b.assign(b, b + L.cout);
    L.raw_b_dev.release();
    LRP_TRY(L.bias.alloc((size_t)L.cout * sizeof(float), total));
    LRP_HIP_CHECK(hipMemcpy(L.bias.p, b, (size_t)L.cout * sizeof(float), hipMemcpyHostToDevice));
    L.have_b = true;
    L.norm_dirty = true;
    return LRP_OK;
  }

  int check_ready() const {
    for (const ConvLayer& L : layers)
      if (!L.have_w || !L.have_b) return fail(LRP_ERR_STATE, "encoder weights for layer '%s' not set", L.name.c_str());
    return LRP_OK;
  }

  // ---- forward once per image -------------------------------------------------------------
  long encode_epoch = 0;   // bumped by every encode: a layer's compact gate is current iff its gc_epoch equals this
  int encode(const float* images_dev, int B, hipStream_t st) {
    if (B < 1 || B > max_images) return fail(LRP_ERR_INVALID, "B=%d outside [1,%d]", B, max_images);
    LRP_TRY(check_ready());
    ++encode_epoch;
    for (ConvLayer& L : layers) L.gfull_epoch = encode_epoch;     // (every path writes the full-resolution gates, except the fused pool below)
    const size_t img_elems = (size_t)img_h * img_w * 3;
    if (gates_pending) {                               // the previous encode's side work still owns G / bufZ / bufXs
      LRP_HIP_CHECK(hipStreamWaitEvent(st, ev_gates, 0));
      gates_pending = false;
    }
    LRP_HIP_CHECK(hipMemsetAsync(act_max.p, 0, act_max.bytes, st));
    LRP_HIP_CHECK(hipMemcpyAsync(images.p, images_dev, B * img_elems * sizeof(float), hipMemcpyDeviceToDevice, st));
    auto im2col_fp32 = [&]() -> int {
      const size_t total = (size_t)B * img_h * img_w * 8;
      hipLaunchKernelGGL(im2col_image_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                         images.as<float>(), a1.as<float>(), B, img_h, img_w);
      LRP_HIP_CHECK(hipGetLastError());
      return LRP_OK;
    };
    float* x = bufX.as<float>();
    float* a = bufA.as<float>();
    float* z = bufZ.as<float>();
    bool mixed = prec == PREC_BF16X3;
    for (const ConvLayer& L : layers)
      if (L.cout & 7) mixed = false;
    const bool overlap = mixed && side && layers.size() > 1;
    std::vector<const float*> xin(layers.size() + 1, nullptr);   // overlapped path: input of every conv
    bool dual = overlap && !fwd_fast;                    // (LRP_PREC_BF16X3_FAST: split-bf16 activation convs + side-stream Z+ chain)
    for (size_t li = 1; li < layers.size(); ++li)
      if (((layers[li].cin | layers[li].cout) & 7) || !layers[li].w_fwd_h.p) dual = false;
    // dual path: the producer hands its consumer the fp16 pairs directly (conv epilogue where no pool follows, the fused
    // pool kernel where one does) — `in_pairs`: pin already holds the operand of the conv about to run
    bool in_pairs = false;
    float *pin = bufXs.as<float>(), *pout = bufXl.as<float>();
    // ... which needs every layer on the interleaved dual matrix (the pairs and the gate leave one epilogue); scales, maxima
    // and unscale records are then kept per IMAGE, so an image's result does not depend on the rest of its batch
    bool emit = dual && fwd_emit() && layers.size() > 1 && layers[0].w_fwd_il.p && !layers[0].pool_after && !(img_elems & 3);
    for (size_t li = 1; li < layers.size(); ++li)
      if (!layers[li].w_fwd_il.p) emit = false;          // (interleaved rows: decided when the weights were packed)
    const size_t per = emit ? (size_t)max_images : 1;    // records per layer
    auto slots_of = [&](size_t lev) { return act_max.as<unsigned>() + lev * per * ACT_MAX_SLOTS; };
    auto unscale_of = [&](size_t li) { return act_unscale.as<float>() + li * per; };
    auto oscale_of = [&](size_t li) { return out_scale.as<float>() + li * per; };
    if (emit)
      for (size_t li = 0; li < layers.size(); ++li) {
        ConvLayer& L = layers[li];
        if (!L.norm_dirty) continue;
        LRP_HIP_CHECK(hipMemsetAsync(L.fnorm.p, 0, 2 * sizeof(float), st));
        // (image layer: the a rows of its 64-wide im2col matrix hold w twice, against x+ and x-: the row sum is 2x the bound)
        hipLaunchKernelGGL(conv_norm_kernel, dim3(L.cout + 1), dim3(256), 0, st, li == 0 ? L.w_fwd.as<float>() : L.w_fwd_a.as<float>(), L.cout,
                           li == 0 ? 64 : 9 * conv_cinp(L.cin), L.bias.as<float>(), L.cout, L.fnorm.as<float>());
        LRP_HIP_CHECK(hipGetLastError());
        L.norm_dirty = false;
      }
    int gate_due = -1;                                 // dual path: layer whose gate waits for the next layer's split (it reads a_l)
    auto launch_gate = [&](int gl) {
      ConvLayer& Lg = layers[gl];
      const size_t ng = (size_t)B * Lg.act_elems();
      hipLaunchKernelGGL(gate_kernel, dim3(stream_grid(ng / 4)), dim3(256), 0, st,
                         reinterpret_cast<const f32x4*>(keep_acts ? Lg.Akeep.as<float>() : Lg.G.as<float>()),
                         reinterpret_cast<const f32x4*>(bufZ.as<float>()), Lg.G.as<f32x4>(), ng / 4);
    };
    for (size_t li = 0; li < layers.size(); ++li) {
      ConvLayer& L = layers[li];
      const bool top = li + 1 == layers.size();
      ConvArgs ca{};
      if (overlap && li > 0) {
        // activation chain only: a_l = relu(conv(x_l) + b) exact fp32, parked in the storage of its future gate
        ca.in = xin[li]; ca.NB = B; ca.H = L.H; ca.W = L.W; ca.Cin = L.cin; ca.CinP = conv_cinp(L.cin); ca.taps = 9;
        ca.bias = L.bias.as<float>(); ca.wpk = L.w_fwd_a.as<float>(); ca.N = L.cout;
        float* a_out = top ? feat.as<float>() : (keep_acts && !L.pool_after) ? L.Akeep.as<float>() : L.G.as<float>();
        ca.out = a_out;
        if (dual) {
          const size_t n8 = (size_t)B * L.H * L.W * L.cin / 8;
          unsigned* slots_in = slots_of(li - 1);
          if (!in_pairs) {
            if (li == 1)
              hipLaunchKernelGGL(absmax_slots_kernel, dim3(stream_grid(n8 * 2)), dim3(256), 0, st, reinterpret_cast<const f32x4*>(xin[li]),
                                 n8 * 2, slots_in);
            hipLaunchKernelGGL(split_h_scaled_kernel, dim3(stream_grid(n8)), dim3(256), 0, st, xin[li], pin, n8, slots_in,
                               unscale_of(li), L.wds.as<float>());
          }
          if (gate_due >= 0) { launch_gate(gate_due); gate_due = -1; }       // a_{l-1} has been read: it may become G_{l-1} now
          LRP_HIP_CHECK(hipGetLastError());
          ConvArgs cd = ca;
          cd.in = pin; cd.wpk = L.w_fwd_h.as<float>(); cd.N = 2 * L.cout; cd.split = L.cout;
          cd.out = a_out; cd.out2 = top ? ztop.as<float>() : bufZ.as<float>();
          cd.in_unscale = unscale_of(li);
          cd.act_max_out = slots_of(li);
          cd.scale_per_img = emit ? 1 : 0; cd.img_rows = L.H * L.W; cd.n_imgs = B;
          cd.dual_il = L.w_fwd_il.p ? 1 : 0;
          const bool fused_gate = cd.dual_il && !top && !L.pool_after;
          if (fused_gate) {
            // a_l and G_l leave the epilogue together: a_l into a ping-pong buffer its consumer (the next layer's split)
            // reads once (or where the fine-tune step looks for it), the gate straight into its cache
            if (!keep_acts) a_out = xin[li] == bufA.as<float>() ? bufX.as<float>() : bufA.as<float>();
            cd.out = a_out; cd.out2 = L.G.as<float>(); cd.dual_gate = 1;
          }
          // pairs for the next conv from THIS layer's epilogue (no pool behind it) or from the fused pool kernel below;
          // their scale comes from a bound that is known now (fwd_scale_kernel), the consumer's unscale with it
          const bool emit_conv = emit && fused_gate, emit_pool = emit && !top && L.pool_after;
          if (emit_conv || emit_pool) {
            hipLaunchKernelGGL(fwd_scale_kernel, dim3(B), dim3(64), 0, st, slots_in, L.fnorm.as<float>(), layers[li + 1].wds.as<float>(),
                               oscale_of(li), unscale_of(li + 1));
            LRP_HIP_CHECK(hipGetLastError());
          }
          if (emit_conv) { cd.pairs_out = pout; cd.pairs_scale = oscale_of(li); cd.skip_out = keep_acts ? 0 : 1; }
          // pooled layer: max-pool, arg-max gate (compact form) and the pooled pairs in THIS conv's epilogue — a_l and Z+_l at
          // full resolution are neither written nor read back (round 4; the pass it replaces: pool_gate_split_kernel below)
          const bool pool_fused = emit_pool && !keep_acts && cd.dual_il && L.Gc.p && L.Gpos.p && conv_takes_pool_fused(L.cout, B, L.H, L.W);
          if (pool_fused) {
            cd.pool_gc = L.Gc.as<float>(); cd.pool_pos = L.Gpos.as<unsigned char>(); cd.pairs_out = pout; cd.pairs_scale = oscale_of(li);
            cd.out = nullptr; cd.out2 = nullptr;
          }
          // The denominators Z+_l of the layers whose reverse launch is two-term (explain(): up to the last pool, >= 576
          // products) are computed two-term as well — with the SAME rounded weights hi(w+) the walk multiplies with.
          // [MI355X: parity at the bench configuration 5.5e-6 -> 4.4e-6, 6 seeds median 4.2e-6 -> 3.3e-6: gate and
          // transposed conv now belong to one (slightly perturbed) network and the rounding largely cancels in R / Z+;
          // two-term Z+ in EVERY layer: 1.0e-4, the top block again.]
          int fterms = 7;
          if (cd.dual_il && walk_f16 && two_term((int)li)) fterms = 23;
          LRP_HIP_CHECK(conv_launch(EPI_FWD_DUAL, cd, st, PREC_F16X2, fterms));
          if (top) break;
          in_pairs = emit_conv || emit_pool;
          if (in_pairs) { float* t = pin; pin = pout; pout = t; }
          if (fused_gate) { xin[li + 1] = a_out; continue; }
          if (L.pool_after && pool_fused) {
            layers[li].gc_epoch = encode_epoch;
            layers[li].gfull_epoch = -1;                 // G itself was not written: expanded on demand (full_gate)
            if (L.idxp.p && L.w_sp.p && sw().sparse_pool) {
              LRP_HIP_CHECK(conv_sparse_index(L.Gpos.as<unsigned char>(), L.idxp.as<unsigned>(), B, L.H / 2, L.W / 2, L.cout, st));
              layers[li].idx_epoch = encode_epoch;
            }
            xin[li + 1] = L.P.as<float>();               // (not read: the next conv takes the pairs)
            continue;
          }
          if (L.pool_after) {
            const size_t n = (size_t)B * L.act_elems();
            if (emit_pool) {
              // pooled activations as pairs (and fp32 only where the fine-tune step looks for them), arg-max gate: one pass
              hipLaunchKernelGGL(pool_gate_split_kernel, dim3(stream_grid(n / 32)), dim3(256), 0, st, L.G.as<float>(), bufZ.as<float>(),
                                 L.G.as<float>(), pin, keep_acts ? L.P.as<float>() : (float*)nullptr, oscale_of(li), B, L.H, L.W, L.cout,
                                 L.Gc.as<float>(), L.Gpos.as<unsigned char>());
              if (L.Gc.p) layers[li].gc_epoch = encode_epoch;
              if (L.idxp.p && L.w_sp.p && sw().sparse_pool) {          // the sparse consumer's index words, once per image
                LRP_HIP_CHECK(conv_sparse_index(L.Gpos.as<unsigned char>(), L.idxp.as<unsigned>(), B, L.H / 2, L.W / 2, L.cout, st));
                layers[li].idx_epoch = encode_epoch;
              }
            } else {
              hipLaunchKernelGGL(maxpool2_kernel, dim3(stream_grid(n / 16)), dim3(256), 0, st, a_out, L.P.as<float>(), B, L.H, L.W, L.cout);
              hipLaunchKernelGGL(pool_gate_kernel, dim3(stream_grid(n / 16)), dim3(256), 0, st, L.G.as<float>(), bufZ.as<float>(),
                                 (float*)nullptr, L.G.as<float>(), B, L.H, L.W, L.cout);
            }
            LRP_HIP_CHECK(hipGetLastError());
            xin[li + 1] = L.P.as<float>();
          } else {
            xin[li + 1] = a_out;
            gate_due = (int)li;
          }
          continue;
        }
        if ((int)li >= fwd_split_from()) {
          // late layers: activation conv in split-bf16 as well (its error passes through few further layers)
          const size_t n8 = (size_t)B * L.H * L.W * L.cin / 8;
          hipLaunchKernelGGL(split_copy_kernel, dim3(stream_grid(n8)), dim3(256), 0, st, xin[li], bufXs.as<float>(), n8);
          LRP_HIP_CHECK(hipGetLastError());
          ca.in = bufXs.as<float>(); ca.wpk = L.w_fwd_as.as<float>();
          LRP_HIP_CHECK(conv_launch(EPI_BIAS_RELU, ca, st, PREC_BF16X3));
        } else {
          LRP_HIP_CHECK(conv_launch(EPI_BIAS_RELU, ca, st));
        }
        if (top) break;
        if (L.pool_after) {
          const size_t n = (size_t)B * L.act_elems();
          hipLaunchKernelGGL(maxpool2_kernel, dim3(stream_grid(n / 16)), dim3(256), 0, st, a_out, L.P.as<float>(), B, L.H, L.W, L.cout);
          LRP_HIP_CHECK(hipGetLastError());
          xin[li + 1] = L.P.as<float>();
        } else {
          xin[li + 1] = a_out;
        }
        continue;
      }
      if (li == 0 && emit) {
        // image layer of the dual forward: the fp32 GEMM over the im2col matrix with interleaved (w | w+-) rows — its epilogue
        // writes the gate G_1, a_1 as the next conv's fp16 pairs (scale from the images' measured maximum) and raises max|a_1|:
        // no gate / absmax / split pass, a_1 itself only where the fine-tune step looks for it
        unsigned* img_slots = slots_of(layers.size());
        hipLaunchKernelGGL(absmax_img_slots_kernel, dim3(64, B), dim3(256), 0, st, images.as<f32x4>(), img_elems / 4, img_slots);
        hipLaunchKernelGGL(fwd_scale_kernel, dim3(B), dim3(64), 0, st, img_slots, L.fnorm.as<float>(), layers[1].wds.as<float>(), oscale_of(0),
                           unscale_of(1));
        // the im2col matrix as fp16 pairs (scaled per image by its own maximum) and the GEMM on the f16 MFMA — unless the
        // fine-tune step is on: its weight gradient of this layer is a product over the fp32 im2col matrix (trainer.h)
        // (the image layer stays on the exact fp32 MFMA: as fp16 pairs it is 0.2-0.4 ms faster per encode and puts the features of
        //  ill-conditioned nets at 1.0e-5 instead of 6.3e-6 — first-layer errors are inherited by every later layer; round 3 kept
        //  that variant behind LRP_FWD_L0_F16, round 4 removed it)
        LRP_TRY(im2col_fp32());
        LRP_HIP_CHECK(hipGetLastError());
        ConvArgs c0{};
        c0.in = a1.as<float>(); c0.NB = B * L.H * L.W; c0.H = 1; c0.W = 1; c0.Cin = 64; c0.CinP = 64; c0.taps = 1;
        c0.bias = L.bias.as<float>(); c0.wpk = L.w_fwd_il.as<float>(); c0.N = 2 * L.cout; c0.split = L.cout;
        c0.in_unscale = unscale_of(0);
        c0.dual_il = 1; c0.dual_gate = 1;
        c0.out = keep_acts ? L.Akeep.as<float>() : nullptr; c0.skip_out = keep_acts ? 0 : 1;
        c0.out2 = L.G.as<float>();
        c0.pairs_out = pin; c0.pairs_scale = oscale_of(0);
        c0.act_max_out = slots_of(0);
        c0.scale_per_img = 1; c0.img_rows = L.H * L.W; c0.n_imgs = B;
        LRP_HIP_CHECK(conv_launch(EPI_FWD_DUAL, c0, st));
        in_pairs = true;
        xin[1] = keep_acts ? L.Akeep.as<float>() : nullptr;
        continue;
      }
      if (li == 0) {
        LRP_TRY(im2col_fp32());
        ca.in = a1.as<float>(); ca.NB = B * L.H * L.W; ca.H = 1; ca.W = 1; ca.Cin = 64; ca.CinP = 64; ca.taps = 1;
      } else {
        ca.in = x; ca.NB = B; ca.H = L.H; ca.W = L.W; ca.Cin = L.cin; ca.CinP = conv_cinp(L.cin); ca.taps = 9;
      }
      ca.bias = L.bias.as<float>();
      float* a_out = top ? feat.as<float>() : a;
      float* z_out = top ? ztop.as<float>() : z;
      if (li > 0 && mixed) {
        // a_l exact (it feeds the next layer), Z+_l in bf16x3 (its error stays inside gate G_l):
        // measured on CPU emulation 4e-6 vs 3e-6 relative L1 for the all-fp32 forward.
        ConvArgs cz = ca;
        ca.wpk = L.w_fwd_a.as<float>(); ca.N = L.cout; ca.out = a_out;
        LRP_HIP_CHECK(conv_launch(EPI_BIAS_RELU, ca, st));
        cz.in = bufXs.as<float>(); cz.wpk = L.w_fwd_zs.as<float>(); cz.N = L.cout; cz.out = z_out;
        LRP_HIP_CHECK(conv_launch(EPI_BIAS, cz, st, PREC_BF16X3));
      } else {
        ca.wpk = L.w_fwd.as<float>();
        ca.N = 2 * L.cout; ca.split = L.cout;
        ca.out = a_out; ca.out2 = z_out;
        LRP_HIP_CHECK(conv_launch(EPI_FWD_DUAL, ca, st));
      }
      if (top) break;
      const size_t n = (size_t)B * L.act_elems();
      if (L.pool_after) {
        hipLaunchKernelGGL(pool_gate_kernel, dim3(stream_grid(n / 16)), dim3(256), 0, st, a, z, x, L.G.as<float>(), B, L.H,
                           L.W, L.cout);
        LRP_HIP_CHECK(hipGetLastError());
        // x now holds pool(a): input of the next conv
      } else {
        hipLaunchKernelGGL(gate_kernel, dim3(stream_grid(n / 4)), dim3(256), 0, st, reinterpret_cast<const f32x4*>(a),
                           reinterpret_cast<const f32x4*>(z), L.G.as<f32x4>(), n / 4);
        LRP_HIP_CHECK(hipGetLastError());
        float* t = x; x = a; a = t;                   // next input = a_l
      }
      xin[li + 1] = x;
      if (keep_acts) {
        // layers that run through the ping-pong buffers (the image layer always; every layer of the exact-fp32 / not
        // overlapped forward): park the next conv's input where layer_input() looks for it
        const size_t bytes = (size_t)B * L.act_elems() * sizeof(float) / (L.pool_after ? 4 : 1);
        LRP_HIP_CHECK(hipMemcpyAsync(L.pool_after ? L.P.p : L.Akeep.p, x, bytes, hipMemcpyDeviceToDevice, st));
      }
      if (mixed && !overlap) {                        // split8 copy of the next conv's input
        const size_t n8 = (size_t)B * layers[li + 1].H * layers[li + 1].W * layers[li + 1].cin / 8;
        hipLaunchKernelGGL(split_copy_kernel, dim3(stream_grid(n8)), dim3(256), 0, st, x, bufXs.as<float>(), n8);
        LRP_HIP_CHECK(hipGetLastError());
      }
    }
    if (dual) {
      if (gate_due >= 0) launch_gate(gate_due);         // (cannot happen for a net that ends in conv layers feeding the top: kept for safety)
      LRP_HIP_CHECK(hipGetLastError());
    } else if (overlap) {
      // side stream, top layer first: layer l's input x_l = a_{l-1} lives in the gate storage of layer l-1, which
      // is turned into G_{l-1} only after layer l is done with it
      LRP_HIP_CHECK(hipEventRecord(ev_fwd, st));
      LRP_HIP_CHECK(hipStreamWaitEvent(side, ev_fwd, 0));
      for (size_t li = layers.size() - 1; li >= 1; --li) {
        ConvLayer& L = layers[li];
        const bool top = li + 1 == layers.size();
        const size_t n8 = (size_t)B * L.H * L.W * L.cin / 8;
        hipLaunchKernelGGL(split_copy_kernel, dim3(stream_grid(n8)), dim3(256), 0, side, xin[li], bufXs.as<float>(), n8);
        LRP_HIP_CHECK(hipGetLastError());
        ConvArgs cz{};
        cz.in = bufXs.as<float>(); cz.NB = B; cz.H = L.H; cz.W = L.W; cz.Cin = L.cin; cz.CinP = conv_cinp(L.cin); cz.taps = 9;
        cz.bias = L.bias.as<float>(); cz.wpk = L.w_fwd_zs.as<float>(); cz.N = L.cout;
        cz.out = top ? ztop.as<float>() : bufZ.as<float>();
        if (!top && !L.pool_after) {
          // no pool behind this layer: G_l = a_l / safe(Z+_l) in the conv's epilogue, Z+_l never written
          cz.gate_src = keep_acts ? L.Akeep.as<float>() : L.G.as<float>();
          cz.out = L.G.as<float>();
          LRP_HIP_CHECK(conv_launch(EPI_BIAS, cz, side, PREC_BF16X3));
          continue;
        }
        LRP_HIP_CHECK(conv_launch(EPI_BIAS, cz, side, PREC_BF16X3));
        if (top) continue;
        const size_t n = (size_t)B * L.act_elems();
        if (L.pool_after)
          hipLaunchKernelGGL(pool_gate_kernel, dim3(stream_grid(n / 16)), dim3(256), 0, side, L.G.as<float>(), bufZ.as<float>(),
                             (float*)nullptr, L.G.as<float>(), B, L.H, L.W, L.cout);
        else
          hipLaunchKernelGGL(gate_kernel, dim3(stream_grid(n / 4)), dim3(256), 0, side,
                             reinterpret_cast<const f32x4*>(keep_acts ? L.Akeep.as<float>() : L.G.as<float>()),
                             reinterpret_cast<const f32x4*>(bufZ.as<float>()), L.G.as<f32x4>(), n / 4);
        LRP_HIP_CHECK(hipGetLastError());
      }
      LRP_HIP_CHECK(hipEventRecord(ev_gates, side));
      gates_pending = true;
    }
    encoded = B;
    features_only = false;
    return LRP_OK;
  }

  // ---- reverse walk, n relevance maps at once ---------------------------------------------
  // R_feat_dev (n, top_h*top_w, top_c) -> R_img_dev (n, img_h, img_w, 3); row2img_dev: device int[n]
  // walk: 0 = LRP (LRPSequentialPresetA); gradient baselines (gradient_based.py:101-265) on the same caches:
  //   1 = Gradient, 2 = InputTimesGradient, 3 = GuidedBackprop — backward-data convs with the full w, the LRP gate
  //   used as the ReLU/arg-max mask, exact fp32.
  // layer_hook (fine-tune step): called with (li, dZ_li) — the gradient at the pre-activation of conv li, n x H x W x cout —
  // before that layer's backward-data conv is launched; the image layer itself is then skipped (R_img_dev may be null).
  // The walks that read a pooled layer's gate at full resolution (EPI_MUL_UP2 / up2_gate: the gradient baselines, the fp32 and
  // fast modes, LRP_UP2_COMPACT=0) after an encode whose fused pool epilogue left the compact form only
  int full_gate(int li, hipStream_t st) {
    ConvLayer& L = layers[li];
    if (!L.pool_after || L.gfull_epoch == encode_epoch) return LRP_OK;
    if (!L.Gc.p || L.gc_epoch != encode_epoch) return fail(LRP_ERR_STATE, "no pool gate of layer %d for this encode", li);
    const size_t n8 = (size_t)encoded * (L.H / 2) * (L.W / 2) * (L.cout / 8);
    hipLaunchKernelGGL(pool_gate_expand_kernel, dim3(stream_grid(n8)), dim3(256), 0, st, L.Gc.as<float>(), L.Gpos.as<unsigned char>(),
                       L.G.as<float>(), encoded, L.H, L.W, L.cout);
    LRP_HIP_CHECK(hipGetLastError());
    L.gfull_epoch = encode_epoch;
    return LRP_OK;
  }

  int explain(int n, const int* row2img_dev, const float* R_feat_dev, float* R_img_dev, hipStream_t st, int walk = 0,
              const std::function<int(int, const float*)>* layer_hook = nullptr) {
    const int* r2i_host = row2img_host;
    row2img_host = nullptr;                              // one-shot
    if (n < 1 || n > max_tokens) return fail(LRP_ERR_INVALID, "n=%d outside [1,%d]", n, max_tokens);
    if (walk < 0 || walk > 3) return fail(LRP_ERR_INVALID, "unknown walk %d", walk);
    if (encoded < 1 || features_only) return fail(LRP_ERR_STATE, "lrp_encode_images must run before the CNN explain");
    if (gates_pending) LRP_HIP_CHECK(hipStreamWaitEvent(st, ev_gates, 0));   // gates / Z_top come from the side stream
    const ConvLayer& T = layers.back();
    float* S = s0.as<float>();
    float* Snext = s1.as<float>();
    bool split = prec == PREC_BF16X3 && walk == 0;
    // Fine-tune step (layer_hook, walk = 1) in the default arithmetic: the backward-data convs run split-bf16 too —
    // bf16 operands (hi + lo), three MFMAs per product, fp32 accumulate; the hook still sees plain fp32 dZ (the weight
    // gradient reads it), so every layer's dZ is written fp32 and re-split by one streaming pass in front of its conv.
    bool hook_split = prec == PREC_BF16X3 && walk == 1 && layer_hook != nullptr;
    for (const ConvLayer& L : layers)
      if (L.cout & 7) split = hook_split = false;       // split8 groups need widths % 8 == 0: exact fp32 otherwise
    for (size_t li = 1; li < layers.size(); ++li)
      if (!layers[li].w_bwd_full_s.p) hook_split = false;
    // LRP_PREC_F16X2: fp16 pairs for S, one fp16 per weight, per-token power-of-two scales (conv_igemm.h PREC_F16X2);
    // needs the fused image layer (the chain's scale is undone in its epilogue)
    const bool f16 = split && walk_f16 && img_fused();
    if (f16) {
      const size_t cnt = (layers.size() + 1) * (size_t)max_tokens;
      if (!tok_exp.p) {
        int64_t dummy = 0;
        LRP_TRY(tok_exp.alloc(cnt * sizeof(int), &dummy));
        LRP_TRY(tok_max.alloc(cnt * sizeof(unsigned), &dummy));
        LRP_TRY(tok_fac.alloc((size_t)max_tokens * sizeof(float), &dummy));
      }
      LRP_HIP_CHECK(hipMemsetAsync(tok_max.p, 0, cnt * sizeof(unsigned), st));
    }
    auto lev_exp = [&](int lev) { return tok_exp.as<int>() + (size_t)lev * max_tokens; };
    auto lev_max = [&](int lev) { return tok_max.as<unsigned>() + (size_t)lev * max_tokens; };
    const int run_prec = f16 ? PREC_F16X2 : (split || hook_split) ? PREC_BF16X3 : PREC_FP32;
    if (walk != 0) {
      const size_t per4 = T.act_elems() / 4;
      hipLaunchKernelGGL(grad_top_kernel, dim3(stream_grid((size_t)n * per4)), dim3(256), 0, st,
                         reinterpret_cast<const f32x4*>(R_feat_dev), feat.as<f32x4>(), row2img_dev,
                         reinterpret_cast<f32x4*>(S), n, per4, walk == 3 ? 1 : 0);
      LRP_HIP_CHECK(hipGetLastError());
    } else if (f16) {
      const int top = (int)layers.size() - 1;
      hipLaunchKernelGGL(top_divide_f16_kernel, dim3(n), dim3(256), 0, st, R_feat_dev, ztop.as<float>(), row2img_dev, S,
                         T.act_elems() / 8, lev_exp(top), lev_max(top));
      LRP_HIP_CHECK(hipGetLastError());
    } else if (split) {
      const size_t per8 = T.act_elems() / 8;
      hipLaunchKernelGGL(top_divide_split_kernel, dim3(stream_grid((size_t)n * per8)), dim3(256), 0, st, R_feat_dev,
                         ztop.as<float>(), row2img_dev, S, n, per8);
      LRP_HIP_CHECK(hipGetLastError());
    } else {
      const size_t per4 = T.act_elems() / 4;
      hipLaunchKernelGGL(top_divide_kernel, dim3(stream_grid((size_t)n * per4)), dim3(256), 0, st,
                         reinterpret_cast<const f32x4*>(R_feat_dev), ztop.as<f32x4>(), row2img_dev,
                         reinterpret_cast<f32x4*>(S), n, per4);
      LRP_HIP_CHECK(hipGetLastError());
    }
    // Compact pool interface (conv_igemm.h ConvArgs::up2_src): where the consumer of a pooled boundary runs on the
    // weights-in-registers kernel (VGG16: block2_conv1 -> pool -> block1_conv2, the 4.1 GB interface), the producer writes
    // its plain fp32 product at POOLED resolution and the consumer builds S = P x gate itself: the 4x-expanded, 75 %-zero
    // tensor is neither written nor read [MI355X, same box: block2_conv1 2.19 -> 1.48 ms (its store stream shrinks 4x, no gate
    // loads), block1_conv2 3.87 -> 4.14 ms (its prologue now multiplies and splits in registers instead of a plain LDS-DMA),
    // walk 26.3 -> 25.7-25.9 ms; heat-map parity unchanged].  LRP_UP2_COMPACT=0 disables.
    const bool up2_on = sw().up2_compact != 0;
    int compact_in = 0;                                  // S (the current layer's input) is in a compact form: 1 fp32 P (BREG consumer), 2 pairs of S_c (pipelined consumer)
    int fold_tw = 0, fold_th = 0;
    // Image layer folded into the epilogue of the layer above it (ConvArgs::img_part): S_1 — 4.1 GB written, 4.5 GB read at
    // the bench configuration — never goes to memory; per tile 160 positions x 6 partial sums do, and a streaming pass
    // adds them up in a fixed order [MI355X, same box: block1_conv2 4.25 -> 4.57 ms (it now also runs the tap GEMM and the
    // in-tile stencil), image layer 1.29 -> 0.18 ms, walk 26.1-26.3 -> 25.3 ms; heat-maps unchanged to fp32 round-off,
    // batch invariance bit-exact].
    // LRP_IMG_FOLD=0 disables.
    const bool fold_on = sw().img_fold != 0;
    for (int li = (int)layers.size() - 1; li >= 0; --li) {
      const ConvLayer& L = layers[li];
      if (layer_hook) {
        LRP_TRY((*layer_hook)(li, S));
        if (li == 0) return LRP_OK;
      }
      ConvArgs ca{};
      ca.in = S; ca.NB = n; ca.H = L.H; ca.W = L.W; ca.Cin = L.cout; ca.CinP = conv_cinp(L.cout); ca.taps = 9;
      if (hook_split) {
        const size_t n8 = (size_t)n * L.act_elems() / 8;
        hipLaunchKernelGGL(split_copy_kernel, dim3(stream_grid(n8)), dim3(256), 0, st, S, bufXs.as<float>(), n8);
        LRP_HIP_CHECK(hipGetLastError());
        ca.in = bufXs.as<float>();
        ca.out_plain = 1;
      }
      ca.wpk = hook_split ? L.w_bwd_full_s.as<float>() : walk != 0 ? L.w_bwd_full.as<float>() : f16 ? L.w_bwd_h.as<float>()
               : split ? L.w_bwd_s.as<float>() : L.w_bwd.as<float>();
      ca.wpk_frag = f16 ? L.w_bwd_frag_h.as<float>() : (split && walk == 0) ? L.w_bwd_frag.as<float>() : nullptr;
      if (f16) {                                        // S_li (level li) -> S_{li-1} (level li - 1); the image layer ends the chain
        hipLaunchKernelGGL(tok_scale_kernel, dim3((n + 255) / 256), dim3(256), 0, st, lev_max(li), lev_exp(li),
                           L.wbs.as<float>(), tok_fac.as<float>(), li > 0 ? lev_exp(li - 1) : (int*)nullptr,
                           n, li == 0 ? 1 : 0);
        LRP_HIP_CHECK(hipGetLastError());
        ca.tok_fac = tok_fac.as<float>();
        if (li > 0) ca.tok_max_out = lev_max(li - 1);
      }
      ca.row2img = row2img_dev;
      ca.order = L.order.get(); ca.row2img_host = r2i_host;
      ca.gate_binary = walk != 0; ca.relu_out = walk == 3;
      int epi;
      if (li == 0 && img_fused()) {
        // T GEMM + 9-tap stencil in one launch (patch tiles, T stays in LDS)
        ca.taps = 1; ca.N = IMG_T_COLS; ca.out = R_img_dev; ca.ximg = images.as<float>();
        ca.img_mode = walk == 0 ? 0 : walk == 2 ? 2 : 1;
        epi = EPI_IMG_STENCIL;
      } else if (li == 0) {
        ca.NB = n * L.H * L.W; ca.H = 1; ca.W = 1; ca.taps = 1;          // 1-tap GEMM over the pixels
        ca.N = IMG_T_COLS; ca.out = Snext; epi = EPI_STORE;
      } else {
        const ConvLayer& P = layers[li - 1];
        ca.N = L.cin; ca.aux = P.G.as<float>(); ca.out = Snext;
        epi = P.pool_after ? EPI_MUL_UP2 : EPI_MUL;
        // the image layer rides on this launch's epilogue?
        if (li == 1 && fold_on && split && !f16 && walk == 0 && !layer_hook && img_fused() && !P.pool_after && L.cin == 64 &&
            P.w_bwd_s.p && conv_takes_breg(L.cin, L.H, L.W, L.w_bwd_frag.p != nullptr)) {
          int hr_ = 0;
          (void)conv_halo_geom(128, L.H, L.W, fold_tw, fold_th, hr_);
          ca.img_w = P.w_bwd_s.as<float>(); ca.img_part = Snext; ca.out = nullptr;
        }
        if (compact_in == 2 && sw().sparse_pool && L.w_sp.p && L.idx_epoch == encode_epoch && !P.pool_after && split && !f16 && walk == 0 &&
            !layer_hook && sp_scp.p) {
          // the pooled boundary on the 2:4-sparse matrix cores (conv_sparse.h): S_c re-laid chunk-major, then one launch per class
          const int Hp = L.H / 2, Wp = L.W / 2;
          const size_t n_sets = (size_t)n * Hp * Wp * (L.cout / 8);
          ProfileRec pr{};
          if (profile) { (void)hipEventCreate(&pr.e0); (void)hipEventCreate(&pr.e1); (void)hipEventRecord(pr.e0, st); }
          hipLaunchKernelGGL(conv_sparse_relayout_kernel, dim3(stream_grid(n_sets)), dim3(256), 0, st, S, sp_scp.as<float>(), n_sets, Hp * Wp, L.cout);
          LRP_HIP_CHECK(hipGetLastError());
          SparseArgs sa{};
          sa.sc = sp_scp.as<float>(); sa.idxp = L.idxp.as<unsigned>(); sa.wsp = L.w_sp.as<float>(); sa.gate = P.G.as<float>(); sa.out = Snext;
          sa.row2img = row2img_dev; sa.NB = n; sa.Hp = Hp; sa.Wp = Wp; sa.C = L.cout; sa.N = L.cin;
          LRP_HIP_CHECK(conv_sparse_launch(sa, st));
          if (profile) {
            (void)hipEventRecord(pr.e1, st);
            pr.flop = 2.0 * (double)n * L.H * L.W * 9.0 * L.cout * L.cin;
            prof.push_back(pr);
          }
          compact_in = 0;
          float* t = S; S = Snext; Snext = t;
          continue;
        }
        if (compact_in == 2 || compact_in == 3) {         // pairs of S_c at pooled resolution (pipelined kernels' loader / the folded BREG launch)
          ca.up2_src = S; ca.up2_pairs = 1; ca.up2_gpos = L.Gpos.as<unsigned char>();
          compact_in = 0;
        } else if (compact_in) {                          // this layer reads the compact form its producer left
          ca.up2_src = S; ca.up2_gate = L.G.as<float>();     // (ca.in = S stays a valid pointer; it is not read)
          const bool gc_on = sw().up2_gc != 0;
          if (gc_on && L.Gc.p && L.gc_epoch == encode_epoch) {   // ... with the gate in compact form too (per-token tiles only)
            ca.up2_gc = L.Gc.as<float>(); ca.up2_gpos = L.Gpos.as<unsigned char>();
          }
          compact_in = 0;
        }
        // does THIS launch write the compact form?  Its consumer is layer li - 1 (N = P.cin, at 2x this resolution)
        if (P.pool_after && up2_on && split && !f16 && walk == 0 && !layer_hook && li >= 2 && P.cin <= 64 && conv_cinp(P.cout) <= 64 &&
            !(P.cout & 7) && conv_takes_breg(P.cin, P.H, P.W, P.w_bwd_frag.p != nullptr)) {
          // pairs mode as below when the consumer will run the folded launch (per-token tiles: its window loader) and this
          // encode left a compact gate; else the plain fp32 product and the consumer multiplies
          const bool gc_on3 = sw().up2_gc != 0 && sw().up2_breg_pairs != 0;
          const bool cons_fold = li == 2 && fold_on && img_fused() && !layers[0].pool_after && P.cin == 64 && layers[0].w_bwd_s.p != nullptr;
          if (gc_on3 && cons_fold && P.Gc.p && P.gc_epoch == encode_epoch) {
            epi = EPI_MUL; ca.aux = P.Gc.as<float>();
            compact_in = 3;
          } else {
            epi = EPI_MUL; ca.gate_none = 1; ca.out_plain = 1;
            compact_in = 1;
          }
        } else if (P.pool_after && up2_on && split && !f16 && walk == 0 && !layer_hook && li >= 2 && !(P.cout & 7) && P.Gc.p &&
                   P.gc_epoch == encode_epoch && conv_takes_pw(P.cin, n, P.H, P.W)) {
          // ... or by a pipelined halo kernel (ConvArgs::up2_pairs): this launch multiplies with the consumer's COMPACT gate
          // (one value per window and channel, at this layer's resolution) and writes S_c as pairs at pooled resolution
          epi = EPI_MUL; ca.aux = P.Gc.as<float>();
          compact_in = 2;
        }
      }
      ProfileRec pr{};
      if (profile) {
        (void)hipEventCreate(&pr.e0); (void)hipEventCreate(&pr.e1);
        (void)hipEventRecord(pr.e0, st);
      }
      // PREC_F16X2: two MFMAs per product (the weights as ONE fp16, 11 bits) below the top block where a sum has at
      // least 576 products (64 channels), three (fp16 pairs on both sides) in the layers after the last pool.  [MI355X, bench configuration, relative L1 vs the float64 graph:
      // all layers three-term 2.8e-6 | two-term up to block4 3.1e-6 | two-term in block5 as well 9.7e-5 — the
      // relevance entering the top block is so concentrated that a sum has one or two dominant products and the
      // weight rounding, the same for every token, no longer averages out; profiles/r02_f16_terms_sweep.txt]
      // Encoder::two_term is that rule; lrp_set_fast_layers replaces it by a per-model mask, LRP_F16_T2MASK (experiments)
      // overrides both.
      const int terms = f16 && two_term(li) ? 5 : 7;
      if (epi == EPI_MUL_UP2) LRP_TRY(full_gate(li - 1, st));
      if (ca.up2_gate) LRP_TRY(full_gate(li, st));
      LRP_HIP_CHECK(conv_launch(epi, ca, st, run_prec, terms));
      if (profile) {
        (void)hipEventRecord(pr.e1, st);
        pr.flop = 2.0 * (double)n * L.H * L.W * 9.0 * L.cout * (li == 0 ? 6 : L.cin);
        prof.push_back(pr);
      }
      if (ca.img_part) {
        // the tiles' partial sums -> R_img (the image layer's second half; its algorithmic flops are booked on this record)
        const ConvLayer& L0 = layers[0];
        ProfileRec p2{};
        if (profile) {
          (void)hipEventCreate(&p2.e0); (void)hipEventCreate(&p2.e1);
          (void)hipEventRecord(p2.e0, st);
        }
        hipLaunchKernelGGL(img_partial_sum_kernel, dim3(stream_grid((size_t)n * L0.H * L0.W)), dim3(256), 0, st, ca.img_part, images.as<float>(),
                           row2img_dev, R_img_dev, n, L0.H, L0.W, fold_th, fold_tw, (L0.W + fold_tw - 1) / fold_tw, 0);
        LRP_HIP_CHECK(hipGetLastError());
        if (profile) {
          (void)hipEventRecord(p2.e1, st);
          p2.flop = 2.0 * (double)n * L0.H * L0.W * 9.0 * L0.cout * 6;
          prof.push_back(p2);
        }
        return LRP_OK;
      }
      float* t = S; S = Snext; Snext = t;
    }
    if (!img_fused()) {  // S now holds T (n, H, W, 54): 9-tap shift-and-add and the x+/x- selection
      const ConvLayer& L0 = layers[0];
      hipLaunchKernelGGL(img_stencil_kernel, dim3(stream_grid((size_t)n * L0.H * L0.W)), dim3(256), 0, st, S,
                         images.as<float>(), row2img_dev, R_img_dev, n, L0.H, L0.W, walk == 0 ? 0 : walk == 2 ? 2 : 1);
      LRP_HIP_CHECK(hipGetLastError());
    }
    return LRP_OK;
  }

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

  int profile_query(int64_t* launches, double* ms, double* flop) {
    int64_t nl = 0; double tm = 0, fl = 0;
    for (ProfileRec& p : prof) {
      float t = 0.f;
      if (hipEventSynchronize(p.e1) == hipSuccess && hipEventElapsedTime(&t, p.e0, p.e1) == hipSuccess) { tm += t; fl += p.flop; ++nl; }
      (void)hipEventDestroy(p.e0); (void)hipEventDestroy(p.e1);
    }
    prof.clear();
    if (launches) *launches = nl;
    if (ms) *ms = tm;
    if (flop) *flop = fl;
    return LRP_OK;
  }
};

}  // namespace lrp
