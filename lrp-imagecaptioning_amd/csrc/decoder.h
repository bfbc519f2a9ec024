// decoder.h — host-side orchestration of the decoder half (state caches, weight
// packing, launch sequences).  Mirrors the reference engine objects
// ExplainImgCaptioningAdaptiveAttention (E:260-666) / ...GridTDModel (E:995-1321):
// forward() == _forward_beam_search, explain() == _explain_lstm_single_word[_sequence].
#pragma once
#include <algorithm>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "cnn_kernels.h"
#include "common.h"
#include "conv_igemm.h"
#include "decoder_gridtd_kernels.h"
#include "decoder_batched_kernels.h"
#include "decoder_kernels.h"
#include "gradient_kernels.h"
#include "train_gemm.h"

namespace lrp {

struct StateBuf {
  DevBuf buf;
  size_t elem_bytes = 4;
};

struct Decoder {
  int kind = 0, L = 0, D = 0, H = 0, E = 0, V = 0, Tm = 0, B_max = 0, NT_max = 0, sos = 2, eos = 1;
  std::map<std::string, std::vector<float>> raw;            // Keras-layout weights until finalize()
  std::map<std::string, std::vector<int64_t>> raw_shape;
  // lrp_set_weight_dev: the Keras-layout matrices stay on the device (no host copy exists); once any weight arrived this
  // way finalize() builds every operand copy with the device packers (repack_device) and uploads host-set stragglers
  std::map<std::string, DevBuf> raw_dev;
  bool finalized = false;
  // packed device weights
  DevBuf w_if_dual, w_v, zero_bias, b_if, Wcat, bcat, Wg, Ws, vvec, Wglob, bglob, Wout, bout, emb, WgT, WglobT, WifT;
  // per-image static part
  DevBuf vfeat, if_pre, ipre, stat, avg, glob_pre;
  // per-step scratch
  DevBuf xh, zgate, hproj, sproj, u, att_pre;
  // cached state (what the reference leaves on `self`)
  DevBuf state_arena;              // declared before its views (destroyed after them)
  std::map<std::string, StateBuf> state;
  DevBuf cap_dev;
  std::vector<int> cap_host, len_host;
  int* cap_pinned = nullptr;       // staging of the captions' H2D copy; ev_cap = that copy (guards reuse without a stream sync)
  hipEvent_t ev_cap = nullptr;
  ~Decoder() {
    if (ev_cap) { (void)hipEventSynchronize(ev_cap); (void)hipEventDestroy(ev_cap); }
    if (cap_pinned) (void)hipHostFree(cap_pinned);
    if (ev_gen) { (void)hipEventSynchronize(ev_gen); (void)hipEventDestroy(ev_gen); }
    if (gen_pinned) (void)hipHostFree(gen_pinned);
  }
  int B_cur = 0;
  bool have_forward = false;
  // explain scratch
  DevBuf rctx, ravg, tailA, w_ifT_pk, w_ifT_pks;
  int prec = PREC_BF16X3;   // arithmetic of the tail GEMM (follows lrp_set_precision)
  // grid-TD only
  DevBuf Wcat2, bcat2, Wg2T, xh1d, xh2d, zg1d, zg2d, hprojd, sprojd, h2u, rho;
  // step-synchronous LRP scan (decoder_batched_kernels.h; allocated and packed on first use)
  bool bx_ready = false;
  DevBuf bxWg1, bxWg2, bx_rc, bx_rh, bx_rglob, bx_q32, bx_acc32;
  DevBuf bx_g[7];          // grid-TD scan state: rc1 rc2 rh1 rh2 rchat nh1 nh2
  // gradient baselines (allocated and packed on first use)
  bool grad_ready = false;
  DevBuf gW1, gW2, gWglob, gWif, g_seed, g_dc1, g_dc2, g_dg, g_out1, g_out2, g_dglob, g_dwords, g_dctx, g_davg, g_tailA;

  int init(const lrp_config& c, int64_t* total) {
    kind = c.decoder; L = c.L; D = c.D; H = c.H; E = c.E; V = c.V; Tm = c.max_caption_len;
    B_max = c.max_images; NT_max = c.max_tokens; sos = c.sos_id; eos = c.eos_id;
    if (L < 1 || D < 4 || H < 4 || E < 4 || V < 2) return fail(LRP_ERR_INVALID, "bad decoder dims");
    if (D % 4 || H % 4) return fail(LRP_ERR_UNSUPPORTED, "D and H must be multiples of 4");
    if (2 * E + H > SCAN_MAXR * 256 || H + 2 * E + H > SCAN_MAXR * 256)
      return fail(LRP_ERR_UNSUPPORTED, "2E+H too large for the scan kernel");
    const size_t B = B_max, S = Tm + 1;
    // every array prepare_static has to zero lives in ONE arena (state_arena): one memset per forward instead of one per
    // array [MI355X, single image: 14 fills of ~4.8 us each on the critical path].  Two passes: sizes, then the views.
    struct Want { std::string nm; size_t bytes, eb; };
    std::vector<Want> wants;
    auto st = [&](const char* nm, size_t elems, size_t eb) -> int {
      wants.push_back({nm, elems * eb, eb});
      return LRP_OK;
    };
    if (kind == LRP_DEC_ADAPTIVE) {
      for (const char* nm : {"ht", "ct", "gt", "it_act", "ft_act", "st", "ot_act"}) LRP_TRY(st(nm, B * S * H, 4));
      LRP_TRY(st("attention", B * S * L, 4));
      LRP_TRY(st("beta", B * S, 4));
      LRP_TRY(st("context", B * S * H, 8));
      LRP_TRY(st("c_hat", B * S * H, 8));
      LRP_TRY(st("xt", B * Tm * 2 * E, 4));
      LRP_TRY(st("caption_preds", B * Tm * V, 8));
    }
    if (kind == LRP_DEC_GRIDTD) {
      for (const char* nm : {"h1t", "c1t", "g1t", "i1t_act", "f1t_act", "h2t", "c2t", "g2t", "i2t_act", "f2t_act",
                             "context", "st", "context_hat", "o1t_act", "o2t_act"})
        LRP_TRY(st(nm, B * S * H, 8));
      LRP_TRY(st("attention", B * S * L, 8));
      LRP_TRY(st("beta", B * S, 8));
      LRP_TRY(st("x1t", B * Tm * (H + 2 * E), 8));
      LRP_TRY(st("x2t", B * Tm * 2 * H, 8));
      LRP_TRY(st("caption_preds", B * Tm * V, 8));
      LRP_TRY(xh1d.alloc(B * (2 * H + 2 * E) * 8, total));
      LRP_TRY(xh2d.alloc(B * 3 * H * 8, total));
      LRP_TRY(zg1d.alloc(B * 5 * H * 8 * KS_GATE, total));
      LRP_TRY(zg2d.alloc(B * 4 * H * 8 * KS_GATE, total));
      LRP_TRY(hprojd.alloc(B * H * 8 * KS_PROJ, total));
      LRP_TRY(sprojd.alloc(B * H * 8 * KS_PROJ, total));
      LRP_TRY(st("#h2u", B * Tm * H, 8));
      LRP_TRY(rho.alloc((size_t)NT_max * Tm * H * 8, total));
    }
    LRP_TRY(vfeat.alloc(B * L * H * 4, total));
    LRP_TRY(if_pre.alloc(B * L * H * 4, total));
    if (kind == LRP_DEC_ADAPTIVE) LRP_TRY(ipre.alloc(B * L * H * 8, total));
    LRP_TRY(stat.alloc(B * L * H * 4, total));
    LRP_TRY(avg.alloc(B * D * 4, total));
    LRP_TRY(glob_pre.alloc(B * E * 4, total));
    LRP_TRY(xh.alloc(B * (2 * E + 2 * H) * 4, total));
    LRP_TRY(zgate.alloc(B * 5 * H * 4 * KS_GATE, total));
    LRP_TRY(hproj.alloc(B * H * 4 * KS_PROJ, total));
    LRP_TRY(sproj.alloc(B * H * 4 * KS_PROJ, total));
    LRP_TRY(att_pre.alloc(B * (L + 1) * 4, total));
    LRP_TRY(st("#u", B * Tm * H, 8));
    {
      size_t off = 0;
      for (const Want& w : wants) off += (w.bytes + 255) / 256 * 256;
      LRP_TRY(state_arena.alloc(off, total));
      off = 0;
      for (const Want& w : wants) {
        if (w.nm == "#u") u.view_of(state_arena, off, w.bytes);
        else if (w.nm == "#h2u") h2u.view_of(state_arena, off, w.bytes);
        else {
          StateBuf& s = state[w.nm];
          s.elem_bytes = w.eb;
          s.buf.view_of(state_arena, off, w.bytes);
        }
        off += (w.bytes + 255) / 256 * 256;
      }
    }
    LRP_TRY(cap_dev.alloc(B * Tm * sizeof(int), total));
    LRP_TRY(rctx.alloc((size_t)NT_max * H * 8, total));
    if ((H & 7) == 0) LRP_TRY(tailA.alloc((size_t)NT_max * L * H * 4, total));
    LRP_TRY(ravg.alloc((size_t)NT_max * D * 8, total));
    cap_host.assign(B * Tm, eos);
    len_host.assign(B, 0);
    return LRP_OK;
  }

  int on_new_features(int B) {
    B_cur = B;
    have_forward = false;
    return LRP_OK;
  }

  bool known_weight(const std::string& nm) const {
    static const char* adaptive_names[] = {"image_features_W", "image_features_b", "global_W", "global_b", "embedding",
                                           "lstm_Wi", "lstm_Wh", "lstm_b", "Wv", "Wg", "V", "Wx", "Wh", "Ws",
                                           "output_W", "output_b"};
    static const char* gridtd_names[] = {"image_features_W", "image_features_b", "global_W", "global_b", "embedding",
                                         "td_Wi", "td_Wh", "td_b", "lang_Wi", "lang_Wh", "lang_b", "W_va", "W_ha", "W_a",
                                         "W_x", "W_h", "W_s", "output_W", "output_b"};
    bool known = false;
    if (kind == LRP_DEC_ADAPTIVE) { for (const char* k : adaptive_names) known |= nm == k; }
    else { for (const char* k : gridtd_names) known |= nm == k; }
    return known;
  }
  int weight_set_common(const std::string& nm, int ndim, const int64_t* shape, size_t* n_out) {
    if (!known_weight(nm)) return fail(LRP_ERR_INVALID, "unknown weight name '%s'", nm.c_str());
    if (raw_stale)
      return fail(LRP_ERR_STATE, "the fine-tune step owns the weights of this handle (read them with lrp_train_get_master)");
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) n *= (size_t)shape[i];
    raw_shape[nm].assign(shape, shape + ndim);
    finalized = false;
    bx_ready = false;                                  // the scan's / gradient path's weight packs are derived too
    grad_ready = false;
    *n_out = n;
    return LRP_OK;
  }
  int set_weight(const std::string& nm, const float* data, int ndim, const int64_t* shape, int64_t*) {
    size_t n = 0;
    LRP_TRY(weight_set_common(nm, ndim, shape, &n));
    raw[nm].assign(data, data + n);
    raw_dev.erase(nm);                                 // (a device copy of the old value, if any, is stale)
    return LRP_OK;
  }
  int set_weight_dev(const std::string& nm, const float* data_dev, int ndim, const int64_t* shape, int64_t* total, hipStream_t st) {
    size_t n = 0;
    LRP_TRY(weight_set_common(nm, ndim, shape, &n));
    DevBuf& d = raw_dev[nm];
    if (d.bytes != n * sizeof(float)) LRP_TRY(d.alloc(n * sizeof(float), total));
    LRP_HIP_CHECK(hipMemcpyAsync(d.p, data_dev, n * sizeof(float), hipMemcpyDeviceToDevice, st));
    raw.erase(nm);
    return LRP_OK;
  }

  // ---- fine-tune step: the derived operand copies rebuilt in place from device weights (adaptive decoder; every
  // buffer below was sized by a previous finalize(); padding stays zero).  Wd(name) = device pointer of the Keras-layout
  // matrix.  Returns 1 when there is nothing to rebuild in place yet (the caller then takes the host path).
  static void tr2d(hipStream_t st, const float* src, int lds, int rows, int cols, float* dst, int ldd) {   // dst[c][r] = src[r][c]
    hipLaunchKernelGGL(dec_transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(32, 8), 0, st, src, lds, rows, cols, dst, ldd);
  }
  static hipError_t cp2d(hipStream_t st, const float* src, int lds, int rows, int cols, float* dst, int ldd) {
    return hipMemcpy2DAsync(dst, (size_t)ldd * 4, src, (size_t)lds * 4, (size_t)cols * 4, rows, hipMemcpyDeviceToDevice, st);
  }
  int refresh_from_device(const std::function<const float*(const char*)>& Wd, hipStream_t st) {
    if (!finalized) return 1;
    if (!raw_stale) {                                  // last moment the set weights are current: build every lazily made pack
      int64_t dummy = 0;
      LRP_TRY(bx_prepare(&dummy, st));
      if (!((H | E | D) & 3)) LRP_TRY(grad_prepare(&dummy, st));
    }
    LRP_TRY(repack_device(Wd, st));
    raw_stale = true;
    return LRP_OK;
  }
  // every derived operand copy (those that exist: the scan / gradient packs only once prepared) from Keras-layout DEVICE
  // matrices: strided D2D copies, tiled transposes, the split kernel.  Buffers must exist with zeroed padding.
  int repack_device(const std::function<const float*(const char*)>& Wd, hipStream_t st) {
    const bool td = kind == LRP_DEC_GRIDTD;
    const float *Wif = Wd("image_features_W"), *Wgl = Wd("global_W");
    const int KD = conv_cinp(D), KH = conv_cinp(H), K4 = conv_cinp(4 * H);
    tr2d(st, Wif, H, D, H, w_if_dual.as<float>(), KD);                                  // rows [0, H) and [H, 2H): W_if^T
    tr2d(st, Wif, H, D, H, w_if_dual.as<float>() + (size_t)H * KD, KD);
    tr2d(st, Wd(td ? "W_va" : "Wv"), H, H, H, w_v.as<float>(), KH);
    tr2d(st, Wgl, E, D, E, WglobT.as<float>(), D);
    tr2d(st, Wif, H, D, H, WifT.as<float>(), D);
    LRP_HIP_CHECK(cp2d(st, Wif, H, D, H, w_ifT_pk.as<float>(), KH));                     // [d][j] = W_if[d][j], K padded
    hipLaunchKernelGGL(split_copy_kernel, dim3(stream_grid(w_ifT_pk.bytes / 32)), dim3(256), 0, st, w_ifT_pk.as<float>(),
                       w_ifT_pks.as<float>(), w_ifT_pk.bytes / 32);
    auto copy = [&](DevBuf& d, const char* nm, size_t n) {
      return hipMemcpyAsync(d.p, Wd(nm), n * 4, hipMemcpyDeviceToDevice, st);
    };
    LRP_HIP_CHECK(copy(b_if, "image_features_b", H)); LRP_HIP_CHECK(copy(Wglob, "global_W", (size_t)D * E));
    LRP_HIP_CHECK(copy(bglob, "global_b", E)); LRP_HIP_CHECK(copy(Wout, "output_W", (size_t)H * V));
    LRP_HIP_CHECK(copy(bout, "output_b", V)); LRP_HIP_CHECK(copy(emb, "embedding", (size_t)V * E));
    LRP_HIP_CHECK(copy(Wg, td ? "W_ha" : "Wg", (size_t)H * H)); LRP_HIP_CHECK(copy(Ws, td ? "W_s" : "Ws", (size_t)H * H));
    LRP_HIP_CHECK(copy(vvec, td ? "W_a" : "V", H));
    // one LSTM: Wcat (Kd x N) = [[Wi | Wsx]; [Wh | Wsh]] (N = 5H with a sentinel, else 4H), bias row, transposed gate-g
    // block WgT (H x Kd), the scan's GEMM operand (rows of the gate-g block, K padded) and the gradient path's pack_rows
    auto lstm = [&](const char* wi, const char* wh, const char* wsx, const char* wsh, const char* bias, int Kx, DevBuf& Wc,
                    DevBuf& bc, DevBuf& WgT_, DevBuf* bxW, DevBuf* gW) -> int {
      const float *Wi = Wd(wi), *Wh = Wd(wh);
      const int Kd = Kx + H, Nn = wsx ? 5 * H : 4 * H;
      float* wc = Wc.as<float>();
      LRP_HIP_CHECK(cp2d(st, Wi, 4 * H, Kx, 4 * H, wc, Nn));
      LRP_HIP_CHECK(cp2d(st, Wh, 4 * H, H, 4 * H, wc + (size_t)Kx * Nn, Nn));
      if (wsx) {
        LRP_HIP_CHECK(cp2d(st, Wd(wsx), H, Kx, H, wc + 4 * H, Nn));
        LRP_HIP_CHECK(cp2d(st, Wd(wsh), H, H, H, wc + (size_t)Kx * Nn + 4 * H, Nn));
      }
      LRP_HIP_CHECK(hipMemcpyAsync(bc.p, Wd(bias), (size_t)4 * H * 4, hipMemcpyDeviceToDevice, st));
      tr2d(st, Wi + 2 * H, 4 * H, Kx, H, WgT_.as<float>(), Kd);
      tr2d(st, Wh + 2 * H, 4 * H, H, H, WgT_.as<float>() + Kx, Kd);
      if (bxW) {
        LRP_HIP_CHECK(cp2d(st, Wi + 2 * H, 4 * H, Kx, H, bxW->as<float>(), KH));
        LRP_HIP_CHECK(cp2d(st, Wh + 2 * H, 4 * H, H, H, bxW->as<float>() + (size_t)Kx * KH, KH));
      }
      if (gW) {                                         // rows = [Wh (H) ; Wi (Kx)], K = 4H
        LRP_HIP_CHECK(cp2d(st, Wh, 4 * H, H, 4 * H, gW->as<float>(), K4));
        LRP_HIP_CHECK(cp2d(st, Wi, 4 * H, Kx, 4 * H, gW->as<float>() + (size_t)H * K4, K4));
      }
      return LRP_OK;
    };
    if (td) {
      LRP_TRY(lstm("td_Wi", "td_Wh", "W_x", "W_h", "td_b", H + 2 * E, Wcat, bcat, WgT, bx_ready ? &bxWg1 : nullptr,
                   grad_ready ? &gW1 : nullptr));
      LRP_TRY(lstm("lang_Wi", "lang_Wh", nullptr, nullptr, "lang_b", 2 * H, Wcat2, bcat2, Wg2T, bx_ready ? &bxWg2 : nullptr,
                   grad_ready ? &gW2 : nullptr));
    } else {
      LRP_TRY(lstm("lstm_Wi", "lstm_Wh", "Wx", "Wh", "lstm_b", 2 * E, Wcat, bcat, WgT, bx_ready ? &bxWg1 : nullptr,
                   grad_ready ? &gW1 : nullptr));
    }
    if (grad_ready) {
      LRP_HIP_CHECK(cp2d(st, Wgl, E, D, E, gWglob.as<float>(), conv_cinp(E)));
      LRP_HIP_CHECK(cp2d(st, Wif, H, D, H, gWif.as<float>(), KH));
    }
    LRP_HIP_CHECK(hipGetLastError());
    return LRP_OK;
  }
  bool raw_stale = false;                                // host copies in `raw` are older than the device operands

  int need(const char* nm, std::initializer_list<int64_t> shp) const {
    auto it = raw_shape.find(nm);
    if (it == raw_shape.end()) return fail(LRP_ERR_STATE, "decoder weight '%s' not set", nm);
    std::vector<int64_t> want(shp);
    // accept (H,1) for vectors given as (H,)
    std::vector<int64_t> got = it->second;
    while (got.size() > 1 && got.back() == 1) got.pop_back();
    while (want.size() > 1 && want.back() == 1) want.pop_back();
    if (got != want) return fail(LRP_ERR_INVALID, "decoder weight '%s' has the wrong shape", nm);
    return LRP_OK;
  }

  static int upload(DevBuf& d, const std::vector<float>& v, int64_t* total) {
    LRP_TRY(d.alloc(v.size() * sizeof(float), total));
    LRP_HIP_CHECK(hipMemcpy(d.p, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
    return LRP_OK;
  }

  // weights shared by both decoders: image_features (dual 1-tap conv), attention projection of the
  // features (Wv / W_va), global feature, embedding, output layer, transposed copies for the LRP tail
  int finalize_common(const char* proj_name, int64_t* total) {
    LRP_TRY(need("image_features_W", {D, H})); LRP_TRY(need("image_features_b", {H}));
    LRP_TRY(need("global_W", {D, E})); LRP_TRY(need("global_b", {E}));
    LRP_TRY(need("embedding", {V, E}));
    LRP_TRY(need(proj_name, {H, H}));
    LRP_TRY(need("output_W", {H, V})); LRP_TRY(need("output_b", {V}));
    const std::vector<float>&Wif = raw["image_features_W"], &Wgl = raw["global_W"];
    std::vector<float> pk;
    {
      const int Np = conv_npad(2 * H), K = conv_cinp(D);
      pk.assign((size_t)Np * K, 0.f);
      pack_conv_fwd(Wif.data(), 1, D, H, 0, Np, pk.data());
      pack_conv_fwd(Wif.data(), 1, D, H, H, Np, pk.data());
      LRP_TRY(upload(w_if_dual, pk, total));
    }
    {
      const int Np = conv_npad(H), K = conv_cinp(H);
      pk.assign((size_t)Np * K, 0.f);
      pack_conv_fwd(raw[proj_name].data(), 1, H, H, 0, Np, pk.data());
      LRP_TRY(upload(w_v, pk, total));
    }
    LRP_TRY(upload(zero_bias, std::vector<float>(std::max(H, E), 0.f), total));
    LRP_TRY(upload(b_if, raw["image_features_b"], total));
    LRP_TRY(upload(Wglob, Wgl, total));
    LRP_TRY(upload(bglob, raw["global_b"], total));
    LRP_TRY(upload(Wout, raw["output_W"], total));
    LRP_TRY(upload(bout, raw["output_b"], total));
    LRP_TRY(upload(emb, raw["embedding"], total));
    pk.assign((size_t)E * D, 0.f);
    for (int d = 0; d < D; ++d)
      for (int e = 0; e < E; ++e) pk[(size_t)e * D + d] = Wgl[(size_t)d * E + e];
    LRP_TRY(upload(WglobT, pk, total));
    pk.assign((size_t)H * D, 0.f);
    for (int d = 0; d < D; ++d)
      for (int j = 0; j < H; ++j) pk[(size_t)j * D + d] = Wif[(size_t)d * H + j];
    LRP_TRY(upload(WifT, pk, total));
    {  // the same matrix as the B operand of the MFMA tail GEMM: out[m][d] = sum_j A[m][j] * W_if[d][j]
      const int Np = conv_npad(D), K = conv_cinp(H);
      std::vector<float> g((size_t)Np * K, 0.f), gs((size_t)Np * K);
      pack_conv_fwd(pk.data(), 1, H, D, 0, Np, g.data());          // pk = WifT [j][d] == "w[ci=j][co=d]"
      pack_split8(g.data(), g.size(), gs.data());
      LRP_TRY(upload(w_ifT_pk, g, total));
      LRP_TRY(upload(w_ifT_pks, gs, total));
    }
    return LRP_OK;
  }

  // [x | h] . [[Wi | Wsent_x] ; [Wh | Wsent_h]] packed (Kx+H) x (4H [+H]) and the transposed gate-g block
  int pack_lstm(const std::vector<float>& Wi, const std::vector<float>& Wh, const std::vector<float>* Wsx,
                const std::vector<float>* Wsh, const std::vector<float>& bias, int Kx, DevBuf& Wc, DevBuf& bc, DevBuf& WgT_,
                int64_t* total) {
    const int Kd = Kx + H, Nn = Wsx ? 5 * H : 4 * H;
    std::vector<float> pk((size_t)Kd * Nn, 0.f);
    for (int k = 0; k < Kd; ++k)
      for (int n = 0; n < Nn; ++n) {
        float v;
        if (k < Kx) v = n < 4 * H ? Wi[(size_t)k * 4 * H + n] : (*Wsx)[(size_t)k * H + n - 4 * H];
        else v = n < 4 * H ? Wh[(size_t)(k - Kx) * 4 * H + n] : (*Wsh)[(size_t)(k - Kx) * H + n - 4 * H];
        pk[(size_t)k * Nn + n] = v;
      }
    LRP_TRY(upload(Wc, pk, total));
    std::vector<float> bcv(Nn, 0.f);
    std::copy(bias.begin(), bias.end(), bcv.begin());
    LRP_TRY(upload(bc, bcv, total));
    pk.assign((size_t)H * Kd, 0.f);                                  // WgT[j][d] = [Wi;Wh][d][2H+j]
    for (int d = 0; d < Kd; ++d)
      for (int j = 0; j < H; ++j)
        pk[(size_t)j * Kd + d] = d < Kx ? Wi[(size_t)d * 4 * H + 2 * H + j] : Wh[(size_t)(d - Kx) * 4 * H + 2 * H + j];
    LRP_TRY(upload(WgT_, pk, total));
    return LRP_OK;
  }

  int finalize_gridtd(int64_t* total) {
    LRP_TRY(finalize_common("W_va", total));
    const int K1 = H + 2 * E;
    LRP_TRY(need("td_Wi", {K1, 4 * H})); LRP_TRY(need("td_Wh", {H, 4 * H})); LRP_TRY(need("td_b", {4 * H}));
    LRP_TRY(need("lang_Wi", {2 * H, 4 * H})); LRP_TRY(need("lang_Wh", {H, 4 * H})); LRP_TRY(need("lang_b", {4 * H}));
    LRP_TRY(need("W_ha", {H, H})); LRP_TRY(need("W_a", {H})); LRP_TRY(need("W_x", {K1, H}));
    LRP_TRY(need("W_h", {H, H})); LRP_TRY(need("W_s", {H, H}));
    LRP_TRY(pack_lstm(raw["td_Wi"], raw["td_Wh"], &raw["W_x"], &raw["W_h"], raw["td_b"], K1, Wcat, bcat, WgT, total));
    LRP_TRY(pack_lstm(raw["lang_Wi"], raw["lang_Wh"], nullptr, nullptr, raw["lang_b"], 2 * H, Wcat2, bcat2, Wg2T, total));
    LRP_TRY(upload(Wg, raw["W_ha"], total));
    LRP_TRY(upload(Ws, raw["W_s"], total));
    LRP_TRY(upload(vvec, raw["W_a"], total));
    finalized = true;
    return LRP_OK;
  }

  // lrp_set_weight_dev path: shapes checked like the host path, operand buffers allocated zeroed, then repack_device
  std::function<const float*(const char*)> raw_dev_lookup() {
    return [this](const char* nm) -> const float* { return raw_dev.at(nm).as<float>(); };
  }
  static int zalloc(DevBuf& d, size_t floats, int64_t* total, hipStream_t st) {
    LRP_TRY(d.alloc(floats * sizeof(float), total));
    LRP_HIP_CHECK(hipMemsetAsync(d.p, 0, d.bytes, st));
    return LRP_OK;
  }
  int finalize_device(int64_t* total, hipStream_t st) {
    const bool td = kind == LRP_DEC_GRIDTD;
    const int K1 = td ? H + 2 * E : 2 * E;              // input width of the (first) LSTM
    LRP_TRY(need("image_features_W", {D, H})); LRP_TRY(need("image_features_b", {H}));
    LRP_TRY(need("global_W", {D, E})); LRP_TRY(need("global_b", {E}));
    LRP_TRY(need("embedding", {V, E})); LRP_TRY(need("output_W", {H, V})); LRP_TRY(need("output_b", {V}));
    if (td) {
      LRP_TRY(need("td_Wi", {K1, 4 * H})); LRP_TRY(need("td_Wh", {H, 4 * H})); LRP_TRY(need("td_b", {4 * H}));
      LRP_TRY(need("lang_Wi", {2 * H, 4 * H})); LRP_TRY(need("lang_Wh", {H, 4 * H})); LRP_TRY(need("lang_b", {4 * H}));
      LRP_TRY(need("W_va", {H, H})); LRP_TRY(need("W_ha", {H, H})); LRP_TRY(need("W_a", {H})); LRP_TRY(need("W_x", {K1, H}));
      LRP_TRY(need("W_h", {H, H})); LRP_TRY(need("W_s", {H, H}));
    } else {
      LRP_TRY(need("lstm_Wi", {K1, 4 * H})); LRP_TRY(need("lstm_Wh", {H, 4 * H})); LRP_TRY(need("lstm_b", {4 * H}));
      LRP_TRY(need("Wv", {H, H})); LRP_TRY(need("Wg", {H, H})); LRP_TRY(need("V", {H}));
      LRP_TRY(need("Wx", {K1, H})); LRP_TRY(need("Wh", {H, H})); LRP_TRY(need("Ws", {H, H}));
    }
    for (auto& kv : raw) {                             // weights that were set from the host: one upload each
      if (raw_dev.count(kv.first)) continue;
      DevBuf& d = raw_dev[kv.first];
      LRP_TRY(d.alloc(kv.second.size() * sizeof(float), total));
      LRP_HIP_CHECK(hipMemcpyAsync(d.p, kv.second.data(), kv.second.size() * sizeof(float), hipMemcpyHostToDevice, st));
    }
    const size_t KD = conv_cinp(D), KH = conv_cinp(H);
    LRP_TRY(zalloc(w_if_dual, (size_t)conv_npad(2 * H) * KD, total, st));
    LRP_TRY(zalloc(w_v, (size_t)conv_npad(H) * KH, total, st));
    LRP_TRY(zalloc(zero_bias, (size_t)std::max(H, E), total, st));
    LRP_TRY(zalloc(b_if, H, total, st)); LRP_TRY(zalloc(Wglob, (size_t)D * E, total, st)); LRP_TRY(zalloc(bglob, E, total, st));
    LRP_TRY(zalloc(Wout, (size_t)H * V, total, st)); LRP_TRY(zalloc(bout, V, total, st)); LRP_TRY(zalloc(emb, (size_t)V * E, total, st));
    LRP_TRY(zalloc(WglobT, (size_t)E * D, total, st)); LRP_TRY(zalloc(WifT, (size_t)H * D, total, st));
    LRP_TRY(zalloc(w_ifT_pk, (size_t)conv_npad(D) * KH, total, st)); LRP_TRY(zalloc(w_ifT_pks, (size_t)conv_npad(D) * KH, total, st));
    LRP_TRY(zalloc(Wcat, (size_t)(K1 + H) * 5 * H, total, st)); LRP_TRY(zalloc(bcat, (size_t)5 * H, total, st));
    LRP_TRY(zalloc(WgT, (size_t)H * (K1 + H), total, st));
    if (td) {
      LRP_TRY(zalloc(Wcat2, (size_t)3 * H * 4 * H, total, st)); LRP_TRY(zalloc(bcat2, (size_t)4 * H, total, st));
      LRP_TRY(zalloc(Wg2T, (size_t)H * 3 * H, total, st));
    }
    LRP_TRY(zalloc(Wg, (size_t)H * H, total, st)); LRP_TRY(zalloc(Ws, (size_t)H * H, total, st)); LRP_TRY(zalloc(vvec, H, total, st));
    finalized = true;
    return repack_device(raw_dev_lookup(), st);
  }
  // (the scan / gradient-path packs in device mode: allocate zeroed, mark ready, let repack_device fill everything)
  int pack_alloc_device(DevBuf& d, int rows, int K, int64_t* total, hipStream_t st) {
    return zalloc(d, (size_t)conv_npad(rows) * conv_cinp(K), total, st);
  }

  int finalize(int64_t* total, hipStream_t st = nullptr) {
    if (finalized) return LRP_OK;
    if (!raw_dev.empty()) return finalize_device(total, st);
    if (kind == LRP_DEC_GRIDTD) return finalize_gridtd(total);
    LRP_TRY(finalize_common("Wv", total));
    LRP_TRY(need("lstm_Wi", {2 * E, 4 * H})); LRP_TRY(need("lstm_Wh", {H, 4 * H})); LRP_TRY(need("lstm_b", {4 * H}));
    LRP_TRY(need("Wg", {H, H})); LRP_TRY(need("V", {H}));
    LRP_TRY(need("Wx", {2 * E, H})); LRP_TRY(need("Wh", {H, H})); LRP_TRY(need("Ws", {H, H}));
    // [x | h_prev] . [[Wi | Wx] ; [Wh | Wh_sentinel]] -> 4H gate pre-activations + H sentinel gate;
    // transposed gate-g block for the LRP scan (E:556-558)
    LRP_TRY(pack_lstm(raw["lstm_Wi"], raw["lstm_Wh"], &raw["Wx"], &raw["Wh"], raw["lstm_b"], 2 * E, Wcat, bcat, WgT, total));
    LRP_TRY(upload(Wg, raw["Wg"], total));
    LRP_TRY(upload(Ws, raw["Ws"], total));
    LRP_TRY(upload(vvec, raw["V"], total));
    finalized = true;
    return LRP_OK;
  }

  // ks > 1: split-K into `ks` slabs of `slab` elements each (the consumer sums them)
  template <typename TX, typename TA, typename TY>
  static hipError_t skinny(const TX* X, int ldx, const float* W, int ldw, const float* bias, TY* Y, int ldy, int R, int K,
                           int N, int relu, hipStream_t st, int ks = 1, size_t slab = 0) {
    int kchunk = ((K + ks - 1) / ks + 63) / 64 * 64;
    const dim3 grid((N + 63) / 64, (R + 31) / 32, ks);
    hipLaunchKernelGGL((skinny_gemm_kernel<TX, TA, TY>), grid, dim3(256), 0, st, X, ldx, W, ldw, bias, Y, ldy, R, K, N, relu,
                       kchunk, slab);
    return hipGetLastError();
  }
  // two products of one shape (R <= 32 rows, no bias) in one launch
  template <typename TX, typename TA, typename TY>
  static hipError_t skinny_pair(const TX* X0, const float* W0, TY* Y0, const TX* X1, const float* W1, TY* Y1, int ldx, int ldw, int ldy,
                                int R, int K, int N, hipStream_t st, int ks, size_t slab) {
    if (R > 32) return hipErrorInvalidValue;
    int kchunk = ((K + ks - 1) / ks + 63) / 64 * 64;
    SkinnyPair p{};
    p.X[0] = X0; p.X[1] = X1; p.W[0] = W0; p.W[1] = W1; p.Y[0] = Y0; p.Y[1] = Y1;
    hipLaunchKernelGGL((skinny_gemm_pair_kernel<TX, TA, TY>), dim3((N + 63) / 64, 2, ks), dim3(256), 0, st, p, ldx, ldw, ldy, R, K, N, kchunk, slab);
    return hipGetLastError();
  }
  static constexpr int KS_GATE = 8, KS_PROJ = 4;

  template <typename T>
  T* S_(const char* nm) { return state[nm].buf.as<T>(); }

  // _forward_beam_search for B images (E:370-436 / E:1092-1178)
  int forward(const float* feat_dev, const int32_t* caps, const int32_t* lens, int B, hipStream_t st) {
    int64_t dummy = 0;
    LRP_TRY(finalize(&dummy, st));
    if (B > B_max) return fail(LRP_ERR_INVALID, "B=%d > max_images=%d", B, B_max);
    int Tmax = 0;
    for (int b = 0; b < B; ++b) {
      if (lens[b] < 1 || lens[b] > Tm) return fail(LRP_ERR_INVALID, "caption %d: length %d outside [1,%d]", b, lens[b], Tm);
      for (int i = 0; i < lens[b]; ++i)
        if (caps[b * Tm + i] < 1 || caps[b * Tm + i] > V)
          return fail(LRP_ERR_INVALID, "caption %d: token id %d outside [1,%d]", b, caps[b * Tm + i], V);
      Tmax = std::max(Tmax, (int)lens[b]);
    }
    if (!cap_pinned) {
      LRP_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&cap_pinned), (size_t)B_max * Tm * sizeof(int)));
      LRP_HIP_CHECK(hipEventCreateWithFlags(&ev_cap, hipEventDisableTiming));
    } else {
      LRP_HIP_CHECK(hipEventSynchronize(ev_cap));        // (the previous call's copy, long done — not the whole stream)
    }
    for (int b = 0; b < B; ++b) {
      len_host[b] = lens[b];
      for (int i = 0; i < Tm; ++i) cap_pinned[b * Tm + i] = cap_host[b * Tm + i] = i < lens[b] ? caps[b * Tm + i] : eos;
    }
    LRP_HIP_CHECK(hipMemcpyAsync(cap_dev.p, cap_pinned, (size_t)B * Tm * sizeof(int), hipMemcpyHostToDevice, st));
    LRP_HIP_CHECK(hipEventRecord(ev_cap, st));
    LRP_TRY(prepare_static(feat_dev, B, st));
    if (kind == LRP_DEC_GRIDTD) return forward_gridtd_steps(B, Tmax, st);
    for (int i = 0; i < Tmax; ++i) LRP_TRY(step_adaptive(i, B, st));
    // ---- output layer for every step at once (E:421-422), float64 like the reference
    LRP_HIP_CHECK((skinny<double, double, double>(u.as<double>(), H, Wout.as<float>(), V, bout.as<float>(),
                                                  S_<double>("caption_preds"), V, B * Tm, H, V, 0, st)));
    B_cur = B;
    have_forward = true;
    return LRP_OK;
  }

  // zeroed state + the per-image static part (E:375-388): relu(F W_if + b), its projection, mean feature, global feature
  int prepare_static(const float* feat_dev, int B, hipStream_t st) {
    LRP_HIP_CHECK(hipMemsetAsync(state_arena.p, 0, state_arena.bytes, st));   // every state array, u and h2u (views of the arena)
    {
      ConvArgs ca{};
      ca.in = feat_dev; ca.NB = B * L; ca.H = 1; ca.W = 1; ca.Cin = D; ca.CinP = conv_cinp(D); ca.taps = 1;
      ca.wpk = w_if_dual.as<float>(); ca.N = 2 * H; ca.split = H; ca.bias = b_if.as<float>();
      ca.out = vfeat.as<float>(); ca.out2 = if_pre.as<float>();
      LRP_HIP_CHECK(conv_launch(EPI_FWD_DUAL, ca, st));
      ConvArgs cv{};
      cv.in = vfeat.as<float>(); cv.NB = B * L; cv.H = 1; cv.W = 1; cv.Cin = H; cv.CinP = conv_cinp(H); cv.taps = 1;
      cv.wpk = w_v.as<float>(); cv.N = H; cv.bias = zero_bias.as<float>(); cv.out = stat.as<float>();
      LRP_HIP_CHECK(conv_launch(EPI_BIAS, cv, st));
    }
    if (kind == LRP_DEC_ADAPTIVE) {
      const size_t ne = (size_t)B * L * H;
      hipLaunchKernelGGL(dec_ipre_kernel, dim3((unsigned)std::min<size_t>((ne + 255) / 256, 2048)), dim3(256), 0, st,
                         if_pre.as<float>(), ipre.as<double>(), ne);
      LRP_HIP_CHECK(hipGetLastError());
    }
    hipLaunchKernelGGL(mean_rows_kernel, dim3(B, (D + 63) / 64), dim3(64), 0, st, feat_dev, avg.as<float>(), L, D);
    LRP_HIP_CHECK(hipGetLastError());
    if (mfma_forward()) {
      // on the fp32 matrix cores like the products of a step (K split over the chip, slices reduced in index order): the
      // VALU GEMV walks K = D alone in 8 workgroups [MI355X, one image: 70 us]
      LRP_TRY(need_sg_ws());
      SgemmArgs a{};
      a.A = avg.as<float>(); a.lda = D; a.B = Wglob.as<float>(); a.ldb = E; a.C = glob_pre.as<float>(); a.ldc = E;
      a.M = B; a.N = E; a.K = D; a.bias = bglob.as<float>();
      LRP_HIP_CHECK(sgemm(a, sg_ws.as<float>(), SG_WS_FLOATS, st));
    } else {
      LRP_HIP_CHECK((skinny<float, float, float>(avg.as<float>(), D, Wglob.as<float>(), E, bglob.as<float>(),
                                                 glob_pre.as<float>(), E, B, D, E, 0, st)));
    }
    return LRP_OK;
  }

  // one decoder step i for B rows (E:399-420): LSTM, attention / sentinel, u[b][i] = h + c_hat
  int step_adaptive(int i, int B, hipStream_t st) {
    const int S = Tm + 1, Kd = 2 * E + H;
    float* ht = S_<float>("ht");
    float* stt = S_<float>("st");
    {
      hipLaunchKernelGGL(dec_prep_x_kernel, dim3(B), dim3(256), 0, st, emb.as<float>(), glob_pre.as<float>(), ht,
                         cap_dev.as<int>(), xh.as<float>(), S_<float>("xt"), i, Tm, E, H, V, sos);
      LRP_HIP_CHECK(hipGetLastError());
      const size_t zslab = (size_t)B_max * 5 * H, pslab = (size_t)B_max * H;
      // the three products of a step on the fp32 matrix cores (train_gemm.h: 64 x 128 tiles, K split over ~240
      // workgroups, slices reduced in index order) — LRP_DEC_MFMA_FWD=0: the VALU skinny GEMM with consumer-side slabs
      const bool mf = mfma_forward();
      int ksg = KS_GATE, ksp = KS_PROJ;
      size_t zsl = zslab, psl = pslab;
      const float *zsrc = zgate.as<float>(), *zbias = nullptr, *hsrc = hproj.as<float>(), *ssrc = sproj.as<float>();
      // MFMA path: the K-split slices stay in the workspace and the CONSUMER kernel adds them up (index order, then the bias
      // — the arithmetic of sgemm_reduce_kernel, so the results do not change): three launches fewer per step.  Regions of
      // the workspace: gates [0, 3/4), h projection [3/4, 7/8), sentinel projection [7/8, 1).
      constexpr size_t WS_G = SG_WS_FLOATS / 4 * 3, WS_P = SG_WS_FLOATS / 8;
      if (mf) {
        LRP_TRY(need_sg_ws());
        SgemmArgs a{};
        a.A = xh.as<float>(); a.lda = Kd; a.B = Wcat.as<float>(); a.ldb = 5 * H; a.M = B; a.N = 5 * H; a.K = Kd;
        LRP_HIP_CHECK(sgemm(a, sg_ws.as<float>(), WS_G, st, &ksg));
        zsrc = sg_ws.as<float>(); zsl = (size_t)B * 5 * H; zbias = bcat.as<float>();
      } else {
        LRP_HIP_CHECK((skinny<float, float, float>(xh.as<float>(), Kd, Wcat.as<float>(), 5 * H, bcat.as<float>(),
                                                   zgate.as<float>(), 5 * H, B, Kd, 5 * H, 0, st, KS_GATE, zslab)));
      }
      hipLaunchKernelGGL(dec_pointwise_kernel, dim3(B, (H + 63) / 64), dim3(64), 0, st, zsrc, ksg, zsl, zbias, ht,
                         S_<float>("ct"), S_<float>("gt"), S_<float>("it_act"), S_<float>("ft_act"), stt,
                         S_<float>("ot_act"), i, Tm, H);
      LRP_HIP_CHECK(hipGetLastError());
      if (mf) {
        SgemmArgs a{};
        a.lda = (long)S * H; a.ldb = H; a.M = B; a.N = H; a.K = H;
        // h . Wg and s . Ws: same shape, ONE launch (the upper half of the grid's z takes the second pair)
        a.A = ht + (size_t)(i + 1) * H; a.B = Wg.as<float>();
        a.A2 = stt + (size_t)(i + 1) * H; a.B2 = Ws.as<float>(); a.ws2 = sg_ws.as<float>() + WS_G + WS_P;
        LRP_HIP_CHECK(sgemm(a, sg_ws.as<float>() + WS_G, WS_P, st, &ksp));
        psl = (size_t)B * H;
        hsrc = sg_ws.as<float>() + WS_G; ssrc = sg_ws.as<float>() + WS_G + WS_P;
      } else {
        LRP_HIP_CHECK((skinny<float, float, float>(ht + (size_t)(i + 1) * H, S * H, Wg.as<float>(), H, nullptr,
                                                   hproj.as<float>(), H, B, H, H, 0, st, KS_PROJ, pslab)));
        LRP_HIP_CHECK((skinny<float, float, float>(stt + (size_t)(i + 1) * H, S * H, Ws.as<float>(), H, nullptr,
                                                   sproj.as<float>(), H, B, H, H, 0, st, KS_PROJ, pslab)));
      }
      hipLaunchKernelGGL(dec_att_scores_kernel, dim3(B, (L + 1 + ATT_ROWS - 1) / ATT_ROWS), dim3(256),
                         (size_t)2 * H * sizeof(float), st, hsrc, ssrc, ksp, psl,
                         stat.as<float>(), vvec.as<float>(), att_pre.as<float>(), L, H);
      LRP_HIP_CHECK(hipGetLastError());
      hipLaunchKernelGGL(dec_att_finish_kernel, dim3(B, (H + 63) / 64), dim3(64), (size_t)(L + 8) * sizeof(float), st,
                         att_pre.as<float>(), if_pre.as<float>(), ht, stt, S_<float>("attention"), S_<float>("beta"),
                         S_<double>("context"), S_<double>("c_hat"), u.as<double>(), i, Tm, L, H);
      LRP_HIP_CHECK(hipGetLastError());
    }
    return LRP_OK;
  }

  // ---- incremental decoding for caption generation: the beam bookkeeping of E:51-120 stays on the host, each search
  // step costs ONE decoder step for all live hypotheses (the reference re-runs the whole captioner per step)
  DevBuf gen_tmp, gen_idx;
  int* gen_pinned = nullptr;
  hipEvent_t ev_gen = nullptr;
  int gen_rows = 0;
  int gen_begin(const float* feat_dev, int B, hipStream_t st) {
    int64_t dummy = 0;
    LRP_TRY(finalize(&dummy, st));
    if (B < 1 || B > B_max) return fail(LRP_ERR_INVALID, "B=%d outside [1,%d]", B, B_max);
    if (!gen_pinned) {
      LRP_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&gen_pinned), (size_t)B_max * 2 * sizeof(int)));
      LRP_HIP_CHECK(hipEventCreateWithFlags(&ev_gen, hipEventDisableTiming));
      LRP_TRY(gen_idx.alloc((size_t)B_max * 2 * sizeof(int), &dummy));
      LRP_TRY(gen_tmp.alloc((size_t)B_max * H * 8, &dummy));
    }
    LRP_HIP_CHECK(hipMemsetAsync(cap_dev.p, 0, cap_dev.bytes, st));
    LRP_TRY(prepare_static(feat_dev, B, st));
    have_forward = false;               // the cached state is a search scratch, not a caption replay
    B_cur = B;
    gen_rows = B;
    return LRP_OK;
  }
  template <typename T>
  int gen_reparent(const char* name, int B, int step, hipStream_t st) {
    T* arr = S_<T>(name);
    hipLaunchKernelGGL(gen_gather_kernel<T>, dim3(B), dim3(256), 0, st, arr, gen_tmp.as<T>(), gen_idx.as<int>(), step, Tm + 1, H);
    hipLaunchKernelGGL(gen_scatter_kernel<T>, dim3(B), dim3(256), 0, st, arr, gen_tmp.as<T>(), step, Tm + 1, H);
    LRP_HIP_CHECK(hipGetLastError());
    return LRP_OK;
  }
  // step s: row r continues row parent[r] with tokenizer id word[r] appended (ignored at s = 0: SOS); logits (B, V) float64
  int gen_step(int B, const int32_t* parent_host, const int32_t* word_host, int s, double* logits_dev, hipStream_t st) {
    if (gen_rows < 1 || have_forward) return fail(LRP_ERR_STATE, "lrp_decoder_gen_begin must run first");
    if (B != gen_rows) return fail(LRP_ERR_INVALID, "B=%d but the search was begun with %d rows", B, gen_rows);
    if (s < 0 || s >= Tm) return fail(LRP_ERR_RANGE, "step %d outside [0,%d)", s, Tm);
    if (s > 0) {
      LRP_HIP_CHECK(hipEventSynchronize(ev_gen));
      for (int r = 0; r < B; ++r) {
        if (parent_host[r] < 0 || parent_host[r] >= B) return fail(LRP_ERR_INVALID, "parent[%d]=%d outside [0,%d)", r, parent_host[r], B);
        if (word_host[r] < 1 || word_host[r] > V) return fail(LRP_ERR_INVALID, "word[%d]=%d outside [1,%d]", r, word_host[r], V);
        gen_pinned[r] = parent_host[r];
        gen_pinned[B + r] = word_host[r];
      }
      LRP_HIP_CHECK(hipMemcpyAsync(gen_idx.p, gen_pinned, (size_t)2 * B * sizeof(int), hipMemcpyHostToDevice, st));
      LRP_HIP_CHECK(hipEventRecord(ev_gen, st));
      if (kind == LRP_DEC_ADAPTIVE) {
        LRP_TRY(gen_reparent<float>("ht", B, s, st));
        LRP_TRY(gen_reparent<float>("ct", B, s, st));
      } else {
        for (const char* nm : {"h1t", "c1t", "h2t", "c2t"}) LRP_TRY(gen_reparent<double>(nm, B, s, st));
      }
      hipLaunchKernelGGL(gen_set_word_kernel, dim3((B + 255) / 256), dim3(256), 0, st, cap_dev.as<int>(), gen_idx.as<int>() + B, s, Tm, B);
      LRP_HIP_CHECK(hipGetLastError());
    }
    LRP_TRY(kind == LRP_DEC_ADAPTIVE ? step_adaptive(s, B, st) : step_gridtd(s, B, st));
    const double* rows = (kind == LRP_DEC_ADAPTIVE ? u.as<double>() : h2u.as<double>()) + (size_t)s * H;
    LRP_HIP_CHECK((skinny<double, double, double>(rows, Tm * H, Wout.as<float>(), V, bout.as<float>(), logits_dev, V, B, H, V, 0, st)));
    return LRP_OK;
  }

  // grid-TD step loop (E:1126-1176): top-down LSTM -> attention/sentinel on h1 -> language LSTM
  int forward_gridtd_steps(int B, int Tmax, hipStream_t st) {
    for (int i = 0; i < Tmax; ++i) LRP_TRY(step_gridtd(i, B, st));
    // logits from h2 alone (E:1154 — the reference quirk), every step at once
    LRP_HIP_CHECK((skinny<double, double, double>(h2u.as<double>(), H, Wout.as<float>(), V, bout.as<float>(),
                                                  S_<double>("caption_preds"), V, B * Tm, H, V, 0, st)));
    B_cur = B;
    have_forward = true;
    return LRP_OK;
  }
  int step_gridtd(int i, int B, hipStream_t st) {
    const int S = Tm + 1, K1 = H + 2 * E;
    double *h1 = S_<double>("h1t"), *h2 = S_<double>("h2t"), *stt = S_<double>("st");
    {
      hipLaunchKernelGGL(gtd_prep_x1_kernel, dim3(B), dim3(256), 0, st, emb.as<float>(), glob_pre.as<float>(), h1, h2,
                         cap_dev.as<int>(), xh1d.as<double>(), S_<double>("x1t"), i, Tm, E, H, V, sos);
      LRP_HIP_CHECK(hipGetLastError());
      const size_t zs1 = (size_t)B_max * 5 * H, zs2 = (size_t)B_max * 4 * H, ps = (size_t)B_max * H;   // split-K slabs
      LRP_HIP_CHECK((skinny<double, double, double>(xh1d.as<double>(), K1 + H, Wcat.as<float>(), 5 * H, bcat.as<float>(),
                                                    zg1d.as<double>(), 5 * H, B, K1 + H, 5 * H, 0, st, KS_GATE, zs1)));
      hipLaunchKernelGGL(gtd_pointwise_kernel, dim3(B), dim3(256), 0, st, zg1d.as<double>(), 5 * H, KS_GATE, zs1, h1, S_<double>("c1t"),
                         S_<double>("g1t"), S_<double>("i1t_act"), S_<double>("f1t_act"), stt, (double*)nullptr,
                         S_<double>("o1t_act"), i, Tm, H);
      LRP_HIP_CHECK(hipGetLastError());
      if (B <= 32) {                                     // h1 . W_ha and s . W_s: one launch
        LRP_HIP_CHECK((skinny_pair<double, double, double>(h1 + (size_t)(i + 1) * H, Wg.as<float>(), hprojd.as<double>(),
                                                           stt + (size_t)(i + 1) * H, Ws.as<float>(), sprojd.as<double>(), S * H, H, H, B, H, H,
                                                           st, KS_PROJ, ps)));
      } else {
        LRP_HIP_CHECK((skinny<double, double, double>(h1 + (size_t)(i + 1) * H, S * H, Wg.as<float>(), H, nullptr,
                                                      hprojd.as<double>(), H, B, H, H, 0, st, KS_PROJ, ps)));
        LRP_HIP_CHECK((skinny<double, double, double>(stt + (size_t)(i + 1) * H, S * H, Ws.as<float>(), H, nullptr,
                                                      sprojd.as<double>(), H, B, H, H, 0, st, KS_PROJ, ps)));
      }
      const size_t lds = (size_t)(2 * H + L + 8) * sizeof(double);
      hipLaunchKernelGGL(gtd_attention_kernel, dim3(B), dim3(256), lds, st, hprojd.as<double>(), sprojd.as<double>(), KS_PROJ, ps,
                         stat.as<float>(), vvec.as<float>(), if_pre.as<float>(), h1, h2, stt, S_<double>("attention"),
                         S_<double>("beta"), S_<double>("context"), S_<double>("context_hat"), xh2d.as<double>(),
                         S_<double>("x2t"), i, Tm, L, H);
      LRP_HIP_CHECK(hipGetLastError());
      LRP_HIP_CHECK((skinny<double, double, double>(xh2d.as<double>(), 3 * H, Wcat2.as<float>(), 4 * H, bcat2.as<float>(),
                                                    zg2d.as<double>(), 4 * H, B, 3 * H, 4 * H, 0, st, KS_GATE, zs2)));
      hipLaunchKernelGGL(gtd_pointwise_kernel, dim3(B), dim3(256), 0, st, zg2d.as<double>(), 4 * H, KS_GATE, zs2, h2, S_<double>("c2t"),
                         S_<double>("g2t"), S_<double>("i2t_act"), S_<double>("f2t_act"), (double*)nullptr,
                         h2u.as<double>(), S_<double>("o2t_act"), i, Tm, H);
      LRP_HIP_CHECK(hipGetLastError());
    }
    return LRP_OK;
  }

  int explain_gridtd(int n, const int* img_dev, const int* t_dev, const int32_t* t_host, const float* feat_dev,
                     float* R_feat_dev, float* att_dev, double* rwords_dev, hipStream_t st) {
    GtdExplainArgs a{};
    a.img_idx = img_dev; a.tpos = t_dev; a.cap = cap_dev.as<int>();
    a.h1t = S_<double>("h1t"); a.c1t = S_<double>("c1t"); a.g1t = S_<double>("g1t"); a.i1t = S_<double>("i1t_act");
    a.f1t = S_<double>("f1t_act"); a.h2t = S_<double>("h2t"); a.c2t = S_<double>("c2t"); a.g2t = S_<double>("g2t");
    a.i2t = S_<double>("i2t_act"); a.f2t = S_<double>("f2t_act"); a.x1t = S_<double>("x1t"); a.x2t = S_<double>("x2t");
    a.ctx = S_<double>("context"); a.st = S_<double>("st"); a.chat = S_<double>("context_hat"); a.beta = S_<double>("beta");
    a.att = S_<double>("attention"); a.preds = S_<double>("caption_preds");
    a.Wout = Wout.as<float>(); a.Wg1T = WgT.as<float>(); a.Wg2T = Wg2T.as<float>(); a.WglobT = WglobT.as<float>();
    a.avg = avg.as<float>(); a.glob_pre = glob_pre.as<float>();
    a.rho = rho.as<double>(); a.ravg = ravg.as<double>();
    a.att_out = att_dev; a.rwords_out = rwords_dev;
    a.Tm = Tm; a.L = L; a.D = D; a.H = H; a.E = E; a.V = V;
    if (batched_scan() && (H & 3) == 0 && t_host) {
      int64_t dummy = 0;
      LRP_TRY(bx_prepare(&dummy, st));
      int t_max = 0;
      for (int i = 0; i < n; ++i) t_max = std::max(t_max, (int)t_host[i]);
      GbxArgs x{};
      x.img_idx = img_dev; x.tpos = t_dev; x.cap = a.cap;
      x.h1t = a.h1t; x.c1t = a.c1t; x.g1t = a.g1t; x.i1t = a.i1t; x.f1t = a.f1t; x.h2t = a.h2t; x.c2t = a.c2t; x.g2t = a.g2t;
      x.i2t = a.i2t; x.f2t = a.f2t; x.x1t = a.x1t; x.x2t = a.x2t; x.ctx = a.ctx; x.st = a.st; x.chat = a.chat; x.beta = a.beta;
      x.att = a.att; x.preds = a.preds; x.Wout = a.Wout; x.WglobT = a.WglobT; x.avg = a.avg; x.glob_pre = a.glob_pre;
      x.rc1 = bx_g[0].as<double>(); x.rc2 = bx_g[1].as<double>(); x.rh1 = bx_g[2].as<double>(); x.rh2 = bx_g[3].as<double>();
      x.rchat = bx_g[4].as<double>(); x.nh1 = bx_g[5].as<double>(); x.nh2 = bx_g[6].as<double>();
      x.rglob = bx_rglob.as<double>(); x.q32 = bx_q32.as<float>(); x.acc32 = bx_acc32.as<float>();
      x.rho = a.rho; x.ravg = a.ravg; x.att_out = att_dev; x.rwords_out = rwords_dev;
      x.Tm = Tm; x.L = L; x.D = D; x.H = H; x.E = E; x.V = V;
      hipLaunchKernelGGL(gbx_head_kernel, dim3(n), dim3(256), 0, st, x);
      LRP_HIP_CHECK(hipGetLastError());
      for (int s = 0; s < t_max; ++s) {
        hipLaunchKernelGGL(gbx_pre2_kernel, dim3(n), dim3(256), 0, st, x, s);
        LRP_HIP_CHECK(hipGetLastError());
        LRP_TRY(gemm_nt(bx_q32.as<float>(), n, H, bxWg2, 3 * H, bx_acc32.as<float>(), st));
        hipLaunchKernelGGL(gbx_mid_kernel, dim3(n), dim3(256), 0, st, x, s);
        LRP_HIP_CHECK(hipGetLastError());
        LRP_TRY(gemm_nt(bx_q32.as<float>(), n, H, bxWg1, 2 * H + 2 * E, bx_acc32.as<float>(), st));
        hipLaunchKernelGGL(gbx_post_kernel, dim3(n), dim3(256), 0, st, x, s);
        LRP_HIP_CHECK(hipGetLastError());
      }
      hipLaunchKernelGGL(gbx_tail_kernel, dim3(n, (D + 63) / 64), dim3(64), (size_t)E * sizeof(double), st, x);
      LRP_HIP_CHECK(hipGetLastError());
    } else {
      const size_t lds = (size_t)(7 * H + std::max(H, E) + E + 8) * sizeof(double);
      hipLaunchKernelGGL(gtd_explain_kernel, dim3(n), dim3(256), lds, st, a);
      LRP_HIP_CHECK(hipGetLastError());
    }
    if (tailA.p && (D & 7) == 0) {
      // tail on the matrix cores, as for the adaptive decoder
      const bool split = prec == PREC_BF16X3 && tail_split();
      hipLaunchKernelGGL(gtd_tail_a_kernel, dim3((L * (H / 8) + 255) / 256, n), dim3(256), 0, st, img_dev, t_dev,
                         if_pre.as<float>(), a.att, a.rho, tailA.as<float>(), Tm, L, H, split ? 1 : 0);
      LRP_HIP_CHECK(hipGetLastError());
      ConvArgs cg{};
      cg.in = tailA.as<float>(); cg.NB = n; cg.H = L; cg.W = 1; cg.Cin = H; cg.CinP = conv_cinp(H); cg.taps = 1;
      cg.wpk = split ? w_ifT_pks.as<float>() : w_ifT_pk.as<float>(); cg.N = D; cg.aux = feat_dev; cg.row2img = img_dev;
      cg.out = R_feat_dev; cg.out_plain = 1;
      LRP_HIP_CHECK(conv_launch(EPI_MUL, cg, st, split ? PREC_BF16X3 : PREC_FP32));
      hipLaunchKernelGGL(tail_finish_kernel, dim3((L * D + 255) / 256 > 64 ? 64 : (L * D + 255) / 256, n), dim3(256), 0, st,
                         img_dev, feat_dev, a.avg, a.ravg, R_feat_dev, L, D);
      LRP_HIP_CHECK(hipGetLastError());
      return LRP_OK;
    }
    GtdTailArgs ta{};
    ta.img_idx = img_dev; ta.tpos = t_dev; ta.F = feat_dev; ta.if_pre = if_pre.as<float>(); ta.att = a.att;
    ta.avg = a.avg; ta.WifT = WifT.as<float>(); ta.rho = a.rho; ta.ravg = a.ravg; ta.R_feat = R_feat_dev;
    ta.Tm = Tm; ta.L = L; ta.D = D; ta.H = H;
    hipLaunchKernelGGL(gtd_tail_kernel, dim3(n, (L + 63) / 64, (D + 63) / 64), dim3(256), 0, st, ta);
    LRP_HIP_CHECK(hipGetLastError());
    return LRP_OK;
  }

  // ---- gradient baselines: _lstm_decoder_backward (E:780-832 adaptive, E:1452-1532 grid-TD), see gradient_kernels.h
  // B operand of a transposed-weight product on conv_igemm: rows = the Keras kernels (in_dim, K) stacked, K padded to 32
  int pack_rows(const std::vector<std::pair<const char*, int>>& blocks, int K, DevBuf& dst, int64_t* total) {
    int N = 0;
    for (auto& b : blocks) N += b.second;
    const int Kp = conv_cinp(K);
    std::vector<float> pk((size_t)conv_npad(N) * Kp, 0.f);
    int r0 = 0;
    for (auto& b : blocks) {
      const std::vector<float>& w = raw.at(b.first);
      if (w.size() != (size_t)b.second * K) return fail(LRP_ERR_INVALID, "weight '%s' has the wrong size for the gradient path", b.first);
      for (int r = 0; r < b.second; ++r) memcpy(&pk[(size_t)(r0 + r) * Kp], &w[(size_t)r * K], (size_t)K * sizeof(float));
      r0 += b.second;
    }
    return upload(dst, pk, total);
  }
  int grad_prepare(int64_t* total, hipStream_t st = nullptr) {
    if (grad_ready) return LRP_OK;
    if ((H & 3) || (E & 3) || (D & 3)) return fail(LRP_ERR_UNSUPPORTED, "gradient path: H, E, D must be multiples of 4");
    const size_t NT = NT_max;
    const bool dev = !raw_dev.empty();                 // weights live on the device: allocate here, repack_device fills
    if (kind == LRP_DEC_ADAPTIVE) {
      if (dev) LRP_TRY(pack_alloc_device(gW1, H + 2 * E, 4 * H, total, st));
      else LRP_TRY(pack_rows({{"lstm_Wh", H}, {"lstm_Wi", 2 * E}}, 4 * H, gW1, total));
      LRP_TRY(g_out1.alloc(NT * (H + 2 * E) * 4, total));
    } else {
      if (dev) {
        LRP_TRY(pack_alloc_device(gW1, 2 * H + 2 * E, 4 * H, total, st));
        LRP_TRY(pack_alloc_device(gW2, 3 * H, 4 * H, total, st));
      } else {
        LRP_TRY(pack_rows({{"td_Wh", H}, {"td_Wi", H + 2 * E}}, 4 * H, gW1, total));
        LRP_TRY(pack_rows({{"lang_Wh", H}, {"lang_Wi", 2 * H}}, 4 * H, gW2, total));
      }
      LRP_TRY(g_out1.alloc(NT * (2 * H + 2 * E) * 4, total));
      LRP_TRY(g_out2.alloc(NT * 3 * H * 4, total));
      LRP_TRY(g_dc2.alloc(NT * H * 4, total));
      LRP_TRY(g_dctx.alloc(NT * Tm * H * 4, total));
    }
    if (dev) {
      LRP_TRY(pack_alloc_device(gWglob, D, E, total, st));
      LRP_TRY(pack_alloc_device(gWif, D, H, total, st));
    } else {
      LRP_TRY(pack_rows({{"global_W", D}}, E, gWglob, total));
      LRP_TRY(pack_rows({{"image_features_W", D}}, H, gWif, total));
    }
    LRP_TRY(g_seed.alloc(NT * H * 4, total));
    LRP_TRY(g_dc1.alloc(NT * H * 4, total));
    LRP_TRY(g_dg.alloc(NT * 4 * H * 4, total));
    LRP_TRY(g_dglob.alloc(NT * E * 4, total));
    LRP_TRY(g_dwords.alloc(NT * Tm * 8, total));
    LRP_TRY(g_davg.alloc(NT * D * 4, total));
    LRP_TRY(g_tailA.alloc(NT * L * H * 4, total));
    grad_ready = true;
    if (dev && finalized) LRP_TRY(repack_device(raw_dev_lookup(), st));
    return LRP_OK;
  }
  // out[M][N] = in[M][K] . W^T   (W packed by pack_rows)
  static int gemm_nt(const float* in, int M, int K, const DevBuf& W, int N, float* out, hipStream_t st) {
    ConvArgs c{};
    c.in = in; c.NB = M; c.H = 1; c.W = 1; c.Cin = K; c.CinP = conv_cinp(K); c.taps = 1; c.N = N;
    c.wpk = W.as<float>(); c.out = out;
    LRP_HIP_CHECK(conv_launch(EPI_STORE, c, st, PREC_FP32));
    return LRP_OK;
  }
  // n units (img_dev[u], t_dev[u]); t_max = largest t among them.  dfeat_dev (n, L, D) fp32, rwords_dev (n, Tm) fp64 or null.
  int gradient(int n, const int* img_dev, const int* t_dev, int t_max, float* dfeat_dev, double* rwords_dev, int64_t* total,
               hipStream_t st) {
    LRP_TRY(grad_prepare(total, st));
    const bool td = kind == LRP_DEC_GRIDTD;
    float* seed = g_seed.as<float>();
    double* dwords = rwords_dev ? rwords_dev : g_dwords.as<double>();
    hipLaunchKernelGGL(grad_seed_kernel, dim3(n), dim3(256), 0, st, img_dev, t_dev, cap_dev.as<int>(), Wout.as<float>(), seed,
                       g_dc1.as<float>(), td ? g_dc2.as<float>() : (float*)nullptr, g_dglob.as<float>(), dwords, Tm, H, E, V);
    LRP_HIP_CHECK(hipGetLastError());
    float* dg = g_dg.as<float>();
    float* o1 = g_out1.as<float>();
    float* o2 = g_out2.as<float>();
    const int N1 = td ? 2 * H + 2 * E : H + 2 * E, N2 = 3 * H;
    for (int s = 0; s < t_max; ++s) {
      if (!td) {
        hipLaunchKernelGGL(grad_cell_kernel<float>, dim3(n), dim3(256), 0, st, img_dev, t_dev, s, seed, o1, N1, 0,
                           (const float*)nullptr, 0, 0, S_<float>("ct"), S_<float>("it_act"), S_<float>("ft_act"),
                           S_<float>("gt"), S_<float>("ot_act"), g_dc1.as<float>(), dg, Tm, H);
        LRP_HIP_CHECK(hipGetLastError());
        LRP_TRY(gemm_nt(dg, n, 4 * H, gW1, N1, o1, st));                  // [d_h[i] | d_x[i] = (words E | glob E)]
        hipLaunchKernelGGL(grad_accum_kernel, dim3(n), dim3(256), 0, st, t_dev, s, o1, N1, H + E, H, g_dglob.as<float>(),
                           dwords, Tm, E);
        LRP_HIP_CHECK(hipGetLastError());
      } else {
        // language LSTM: d_h2[i+1] = seed (s = 0) | carried d_h2[i+1] + d_x1[i+1][:H] (E:1522)
        hipLaunchKernelGGL(grad_cell_kernel<double>, dim3(n), dim3(256), 0, st, img_dev, t_dev, s, seed, o2, N2, 0,
                           s > 0 ? o1 : (const float*)nullptr, N1, H, S_<double>("c2t"), S_<double>("i2t_act"),
                           S_<double>("f2t_act"), S_<double>("g2t"), S_<double>("o2t_act"), g_dc2.as<float>(), dg, Tm, H);
        LRP_HIP_CHECK(hipGetLastError());
        LRP_TRY(gemm_nt(dg, n, 4 * H, gW2, N2, o2, st));                  // [d_h2[i] | d_x2 = (c_hat H | h1 H)]
        hipLaunchKernelGGL(gtd_grad_ctx_kernel, dim3(n), dim3(256), 0, st, img_dev, t_dev, s, seed, o2, N2, H,
                           S_<double>("beta"), g_dctx.as<float>(), Tm, H);
        LRP_HIP_CHECK(hipGetLastError());
        // top-down LSTM: d_h1[i+1] = carried d_h1[i+1] + d_x2[H:] (E:1504)
        hipLaunchKernelGGL(grad_cell_kernel<double>, dim3(n), dim3(256), 0, st, img_dev, t_dev, s, (const float*)nullptr, o1, N1,
                           0, o2, N2, 2 * H, S_<double>("c1t"), S_<double>("i1t_act"), S_<double>("f1t_act"),
                           S_<double>("g1t"), S_<double>("o1t_act"), g_dc1.as<float>(), dg, Tm, H);
        LRP_HIP_CHECK(hipGetLastError());
        LRP_TRY(gemm_nt(dg, n, 4 * H, gW1, N1, o1, st));                  // [d_h1[i] | d_x1 = (h2 H | glob E | words E)]
        hipLaunchKernelGGL(grad_accum_kernel, dim3(n), dim3(256), 0, st, t_dev, s, o1, N1, 2 * H, 2 * H + E,
                           g_dglob.as<float>(), dwords, Tm, E);
        LRP_HIP_CHECK(hipGetLastError());
      }
    }
    hipLaunchKernelGGL(grad_glob_mask_kernel, dim3(n), dim3(256), 0, st, img_dev, glob_pre.as<float>(), g_dglob.as<float>(), E,
                       td ? 0 : 1);
    LRP_HIP_CHECK(hipGetLastError());
    LRP_TRY(gemm_nt(g_dglob.as<float>(), n, E, gWglob, D, g_davg.as<float>(), st));
    const dim3 tg((unsigned)std::min((L * H + 255) / 256, 64), n);
    if (td)
      hipLaunchKernelGGL(grad_tail_a_kernel<double>, tg, dim3(256), 0, st, img_dev, t_dev, seed, g_dctx.as<float>(),
                         S_<double>("attention"), if_pre.as<float>(), g_tailA.as<float>(), Tm, L, H);
    else
      hipLaunchKernelGGL(grad_tail_a_kernel<float>, tg, dim3(256), 0, st, img_dev, t_dev, seed, (const float*)nullptr,
                         S_<float>("attention"), if_pre.as<float>(), g_tailA.as<float>(), Tm, L, H);
    LRP_HIP_CHECK(hipGetLastError());
    LRP_TRY(gemm_nt(g_tailA.as<float>(), n * L, H, gWif, D, dfeat_dev, st));
    hipLaunchKernelGGL(grad_tail_finish_kernel, dim3((unsigned)std::min((L * D + 255) / 256, 64), n), dim3(256), 0, st,
                       g_davg.as<float>(), dfeat_dev, L, D);
    LRP_HIP_CHECK(hipGetLastError());
    return LRP_OK;
  }

  int check_token(int b, int t) const {
    if (!have_forward) return fail(LRP_ERR_STATE, "lrp_decoder_forward must run before explain");
    if (b < 0 || b >= B_cur) return fail(LRP_ERR_INVALID, "image index %d outside the %d forwarded captions", b, B_cur);
    if (t < 1 || t > len_host[b]) return fail(LRP_ERR_RANGE, "index out of range of captions (t=%d, caption length %d)", t, len_host[b]);
    return LRP_OK;
  }

  // B operand [rows][K] (fp32, padded like pack_rows) from a host matrix
  int pack_matrix(const std::vector<float>& m, int N, int K, DevBuf& dst, int64_t* total) {
    const int Kp = conv_cinp(K);
    std::vector<float> pk((size_t)conv_npad(N) * Kp, 0.f);
    for (int r = 0; r < N; ++r) memcpy(&pk[(size_t)r * Kp], &m[(size_t)r * K], (size_t)K * sizeof(float));
    return upload(dst, pk, total);
  }
  // gate-g block of an LSTM, [Wi;Wh][:, 2H:3H] (E:556-558), as (Kx + H) rows x H
  std::vector<float> gate_g_block(const char* wi, const char* wh, int Kx) const {
    const std::vector<float>& Wi = raw.at(wi);
    const std::vector<float>& Wh = raw.at(wh);
    std::vector<float> g((size_t)(Kx + H) * H);
    for (int d = 0; d < Kx + H; ++d)
      for (int j = 0; j < H; ++j)
        g[(size_t)d * H + j] = d < Kx ? Wi[(size_t)d * 4 * H + 2 * H + j] : Wh[(size_t)(d - Kx) * 4 * H + 2 * H + j];
    return g;
  }
  // tail GEMM of the decoder LRP (R at the image_features layer back to the CNN features): exact fp32 MFMA — its split-bf16 form
  // saved 0.2 ms of a 40 ms step and cost the decoder half 5x of its parity margin (R_feat 3-10e-6 instead of 0.4-1.2e-6 vs
  // the float64 oracle); it was kept behind LRP_DEC_TAIL_SPLIT until round 4.
  static constexpr bool tail_split() { return false; }
  DevBuf sg_ws;                                        // K-split partials of the forward's matrix-core products
  static constexpr size_t SG_WS_FLOATS = (size_t)4 << 20;
  int need_sg_ws() {
    if (!sg_ws.p) { int64_t dummy = 0; LRP_TRY(sg_ws.alloc(SG_WS_FLOATS * sizeof(float), &dummy)); }
    return LRP_OK;
  }
  static bool mfma_forward() { return sw().dec_mfma_fwd != 0; }   // LRP_DEC_MFMA_FWD=0: VALU skinny GEMMs in the forward
  static bool batched_scan() { return sw().dec_batched != 0; }    // LRP_DEC_BATCHED=0: one workgroup per unit (dec_explain_adaptive_kernel)
  int bx_prepare(int64_t* total, hipStream_t st = nullptr) {
    if (bx_ready) return LRP_OK;
    const size_t NT = NT_max;
    const bool dev = !raw_dev.empty();
    if (kind == LRP_DEC_ADAPTIVE) {
      if (dev) LRP_TRY(pack_alloc_device(bxWg1, 2 * E + H, H, total, st));
      else LRP_TRY(pack_matrix(gate_g_block("lstm_Wi", "lstm_Wh", 2 * E), 2 * E + H, H, bxWg1, total));
      LRP_TRY(bx_acc32.alloc(NT * (2 * E + H) * 4, total));
    } else {
      if (dev) {
        LRP_TRY(pack_alloc_device(bxWg1, 2 * H + 2 * E, H, total, st));
        LRP_TRY(pack_alloc_device(bxWg2, 3 * H, H, total, st));
      } else {
        LRP_TRY(pack_matrix(gate_g_block("td_Wi", "td_Wh", H + 2 * E), 2 * H + 2 * E, H, bxWg1, total));
        LRP_TRY(pack_matrix(gate_g_block("lang_Wi", "lang_Wh", 2 * H), 3 * H, H, bxWg2, total));
      }
      LRP_TRY(bx_acc32.alloc(NT * (size_t)std::max(2 * H + 2 * E, 3 * H) * 4, total));
      for (DevBuf& d : bx_g) LRP_TRY(d.alloc(NT * H * 8, total));
    }
    LRP_TRY(bx_rc.alloc(NT * H * 8, total));
    LRP_TRY(bx_rh.alloc(NT * H * 8, total));
    LRP_TRY(bx_rglob.alloc(NT * E * 8, total));
    LRP_TRY(bx_q32.alloc(NT * H * 4, total));
    bx_ready = true;
    if (dev && finalized) LRP_TRY(repack_device(raw_dev_lookup(), st));
    return LRP_OK;
  }

  int explain(int n, const int* img_dev, const int* t_dev, const int32_t*, const int32_t* t_host, int variant,
              const float* feat_dev, float* R_feat_dev, float* att_dev, double* rwords_dev, hipStream_t st) {
    if (variant != LRP_EXPLAIN_SEQUENCE && variant != LRP_EXPLAIN_SINGLE_STEP) return fail(LRP_ERR_INVALID, "bad variant");
    if (kind == LRP_DEC_GRIDTD) {
      if (variant != LRP_EXPLAIN_SEQUENCE)      // E:167-172: the grid-TD class does not override _explain_lstm_single_word
        return fail(LRP_ERR_UNSUPPORTED, "the grid-TD decoder has no single-step variant");
      return explain_gridtd(n, img_dev, t_dev, t_host, feat_dev, R_feat_dev, att_dev, rwords_dev, st);
    }
    ExplainArgs a{};
    a.img_idx = img_dev; a.tpos = t_dev; a.cap = cap_dev.as<int>();
    a.ht = S_<float>("ht"); a.ct = S_<float>("ct"); a.gt = S_<float>("gt"); a.it = S_<float>("it_act");
    a.ft = S_<float>("ft_act"); a.st = S_<float>("st"); a.beta = S_<float>("beta"); a.att = S_<float>("attention");
    a.xt = S_<float>("xt"); a.ctx = S_<double>("context"); a.chat = S_<double>("c_hat");
    a.preds = S_<double>("caption_preds");
    a.Wout = Wout.as<float>(); a.WgT = WgT.as<float>(); a.WglobT = WglobT.as<float>();
    a.avg = avg.as<float>(); a.glob_pre = glob_pre.as<float>();
    a.rctx = rctx.as<double>(); a.ravg = ravg.as<double>();
    a.att_out = att_dev; a.rwords_out = rwords_dev;
    a.Tm = Tm; a.L = L; a.D = D; a.H = H; a.E = E; a.V = V; a.single_step = variant == LRP_EXPLAIN_SINGLE_STEP;
    if (batched_scan() && (H & 3) == 0 && t_host) {
      // step-synchronous scan: per step one pointwise kernel, ONE GEMM over all units, one routing kernel
      int64_t dummy = 0;
      LRP_TRY(bx_prepare(&dummy, st));
      int t_max = 0;
      for (int i = 0; i < n; ++i) t_max = std::max(t_max, (int)t_host[i]);
      BxArgs x{};
      x.img_idx = img_dev; x.tpos = t_dev; x.cap = a.cap; x.ht = a.ht; x.ct = a.ct; x.gt = a.gt; x.it = a.it; x.ft = a.ft;
      x.st = a.st; x.beta = a.beta; x.att = a.att; x.xt = a.xt; x.ctx = a.ctx; x.chat = a.chat; x.preds = a.preds;
      x.Wout = a.Wout; x.WglobT = a.WglobT; x.avg = a.avg; x.glob_pre = a.glob_pre;
      x.rc = bx_rc.as<double>(); x.rh = bx_rh.as<double>(); x.rglob = bx_rglob.as<double>(); x.q32 = bx_q32.as<float>();
      x.acc32 = bx_acc32.as<float>(); x.rctx = a.rctx; x.ravg = a.ravg; x.att_out = att_dev; x.rwords_out = rwords_dev;
      x.Tm = Tm; x.L = L; x.D = D; x.H = H; x.E = E; x.V = V; x.single_step = a.single_step;
      hipLaunchKernelGGL(bx_head_kernel, dim3(n), dim3(256), 0, st, x);
      LRP_HIP_CHECK(hipGetLastError());
      const int steps = a.single_step ? 1 : t_max;
      hipLaunchKernelGGL(bx_pre_kernel, dim3(n), dim3(256), 0, st, x, 0);
      LRP_HIP_CHECK(hipGetLastError());
      for (int s = 0; s < steps; ++s) {
        LRP_TRY(gemm_nt(bx_q32.as<float>(), n, H, bxWg1, 2 * E + H, bx_acc32.as<float>(), st));
        hipLaunchKernelGGL(bx_post_kernel, dim3(n), dim3(256), 0, st, x, s, s + 1 < steps ? 1 : 0);   // post(s) + pre(s + 1): one launch
        LRP_HIP_CHECK(hipGetLastError());
      }
      hipLaunchKernelGGL(bx_tail_kernel, dim3(n, (D + 63) / 64), dim3(64), (size_t)E * sizeof(double), st, x);
      LRP_HIP_CHECK(hipGetLastError());
    } else {
      const size_t lds = (size_t)(2 * H + std::max(H, E) + E + 8) * sizeof(double);
      hipLaunchKernelGGL(dec_explain_adaptive_kernel, dim3(n), dim3(256), lds, st, a);
      LRP_HIP_CHECK(hipGetLastError());
    }
    if (tailA.p && (D & 7) == 0) {
      // tail on the matrix cores: A operand -> 1-tap conv_igemm with the F-multiply as its gate -> mean-pool share
      const bool split = prec == PREC_BF16X3 && tail_split();
      TailAArgs aa{};
      aa.img_idx = img_dev; aa.tpos = t_dev; aa.vfeat = vfeat.as<float>(); aa.ipre = ipre.as<double>(); aa.att = a.att;
      aa.rho = a.rctx; aa.A = tailA.as<float>(); aa.Tm = Tm; aa.L = L; aa.H = H; aa.split = split;
      hipLaunchKernelGGL(tail_a_kernel, dim3((L * (H / 8) + 255) / 256, n), dim3(256), 0, st, aa);
      LRP_HIP_CHECK(hipGetLastError());
      ConvArgs cg{};
      cg.in = tailA.as<float>(); cg.NB = n; cg.H = L; cg.W = 1; cg.Cin = H; cg.CinP = conv_cinp(H); cg.taps = 1;
      cg.wpk = split ? w_ifT_pks.as<float>() : w_ifT_pk.as<float>(); cg.N = D; cg.aux = feat_dev; cg.row2img = img_dev;
      cg.out = R_feat_dev; cg.out_plain = 1;
      LRP_HIP_CHECK(conv_launch(EPI_MUL, cg, st, split ? PREC_BF16X3 : PREC_FP32));
      hipLaunchKernelGGL(tail_finish_kernel, dim3((L * D + 255) / 256 > 64 ? 64 : (L * D + 255) / 256, n), dim3(256), 0, st,
                         img_dev, feat_dev, a.avg, a.ravg, R_feat_dev, L, D);
      LRP_HIP_CHECK(hipGetLastError());
      return LRP_OK;
    }
    TailArgs ta{};
    ta.img_idx = img_dev; ta.tpos = t_dev; ta.F = feat_dev; ta.vfeat = vfeat.as<float>(); ta.ipre = ipre.as<double>();
    ta.att = a.att;
    ta.avg = a.avg; ta.WifT = WifT.as<float>(); ta.rctx = a.rctx; ta.ravg = a.ravg; ta.R_feat = R_feat_dev;
    ta.Tm = Tm; ta.L = L; ta.D = D; ta.H = H;
    hipLaunchKernelGGL(dec_tail_kernel, dim3(n, (L + 63) / 64, (D + 63) / 64), dim3(256), 0, st, ta);
    LRP_HIP_CHECK(hipGetLastError());
    return LRP_OK;
  }

  int read_state(const char* name, void* out_dev, size_t out_bytes, hipStream_t st) {
    std::string nm(name);
    const DevBuf* src = nullptr;
    auto it = state.find(nm);
    if (it != state.end()) src = &it->second.buf;
    else if (nm == "image_features_before_act") src = &if_pre;
    else if (nm == "average_img_feature") src = &avg;
    else if (nm == "global_img_feature_before_act") src = &glob_pre;
    else if (nm == "total_static_img_feature" || nm == "image_features_proj") src = &stat;
    if (!src) return fail(LRP_ERR_INVALID, "unknown state array '%s'", name);
    if (out_bytes > src->bytes) return fail(LRP_ERR_INVALID, "state '%s' holds %zu bytes, %zu requested", name, src->bytes, out_bytes);
    LRP_HIP_CHECK(hipMemcpyAsync(out_dev, src->p, out_bytes, hipMemcpyDeviceToDevice, st));
    return LRP_OK;
  }
};

}  // namespace lrp
