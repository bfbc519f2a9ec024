// decoder.h — host-side orchestration of the decoder half (state caches, weight
// packing, launch sequences).  Mirrors the reference engine objects
// ExplainImgCaptioningAdaptiveAttention (E:260-666) / ...GridTDModel (E:995-1321):
// forward() == _forward_beam_search, explain() == _explain_lstm_single_word[_sequence].
#pragma once
#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "common.h"
#include "conv_igemm.h"
#include "decoder_kernels.h"

namespace lrp {

struct StateBuf {
  DevBuf buf;
  size_t elem_bytes = 4;
};

struct Decoder {
  int kind = 0, L = 0, D = 0, H = 0, E = 0, V = 0, Tm = 0, B_max = 0, NT_max = 0, sos = 2, eos = 1;
  std::map<std::string, std::vector<float>> raw;            // Keras-layout weights until finalize()
  std::map<std::string, std::vector<int64_t>> raw_shape;
  bool finalized = false;
  // packed device weights
  DevBuf w_if_dual, w_v, zero_bias, b_if, Wcat, bcat, Wg, Ws, vvec, Wglob, bglob, Wout, bout, emb, WgT, WglobT, WifT;
  // per-image static part
  DevBuf vfeat, if_pre, stat, avg, glob_pre;
  // per-step scratch
  DevBuf xh, zgate, hproj, sproj, u;
  // cached state (what the reference leaves on `self`)
  std::map<std::string, StateBuf> state;
  DevBuf cap_dev;
  std::vector<int> cap_host, len_host;
  int B_cur = 0;
  bool have_forward = false;
  // explain scratch
  DevBuf rctx, ravg;

  int init(const lrp_config& c, int64_t* total) {
    kind = c.decoder; L = c.L; D = c.D; H = c.H; E = c.E; V = c.V; Tm = c.max_caption_len;
    B_max = c.max_images; NT_max = c.max_tokens; sos = c.sos_id; eos = c.eos_id;
    if (L < 1 || D < 4 || H < 4 || E < 4 || V < 2) return fail(LRP_ERR_INVALID, "bad decoder dims");
    if (D % 4 || H % 4) return fail(LRP_ERR_UNSUPPORTED, "D and H must be multiples of 4");
    if (2 * E + H > SCAN_MAXR * 256 || H + 2 * E + H > SCAN_MAXR * 256)
      return fail(LRP_ERR_UNSUPPORTED, "2E+H too large for the scan kernel");
    const size_t B = B_max, S = Tm + 1;
    auto st = [&](const char* nm, size_t elems, size_t eb) -> int {
      StateBuf& s = state[nm];
      s.elem_bytes = eb;
      return s.buf.alloc(elems * eb, total);
    };
    if (kind == LRP_DEC_ADAPTIVE) {
      for (const char* nm : {"ht", "ct", "gt", "it_act", "ft_act", "st"}) LRP_TRY(st(nm, B * S * H, 4));
      LRP_TRY(st("attention", B * S * L, 4));
      LRP_TRY(st("beta", B * S, 4));
      LRP_TRY(st("context", B * S * H, 8));
      LRP_TRY(st("c_hat", B * S * H, 8));
      LRP_TRY(st("xt", B * Tm * 2 * E, 4));
      LRP_TRY(st("caption_preds", B * Tm * V, 8));
    }
    LRP_TRY(vfeat.alloc(B * L * H * 4, total));
    LRP_TRY(if_pre.alloc(B * L * H * 4, total));
    LRP_TRY(stat.alloc(B * L * H * 4, total));
    LRP_TRY(avg.alloc(B * D * 4, total));
    LRP_TRY(glob_pre.alloc(B * E * 4, total));
    LRP_TRY(xh.alloc(B * (2 * E + 2 * H) * 4, total));
    LRP_TRY(zgate.alloc(B * 5 * H * 4, total));
    LRP_TRY(hproj.alloc(B * H * 4, total));
    LRP_TRY(sproj.alloc(B * H * 4, total));
    LRP_TRY(u.alloc(B * Tm * H * 8, total));
    LRP_TRY(cap_dev.alloc(B * Tm * sizeof(int), total));
    LRP_TRY(rctx.alloc((size_t)NT_max * H * 8, total));
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

  int set_weight(const std::string& nm, const float* data, int ndim, const int64_t* shape, int64_t*) {
    static const char* adaptive_names[] = {"image_features_W", "image_features_b", "global_W", "global_b", "embedding",
                                           "lstm_Wi", "lstm_Wh", "lstm_b", "Wv", "Wg", "V", "Wx", "Wh", "Ws",
                                           "output_W", "output_b"};
    static const char* gridtd_names[] = {"image_features_W", "image_features_b", "global_W", "global_b", "embedding",
                                         "td_Wi", "td_Wh", "td_b", "lang_Wi", "lang_Wh", "lang_b", "W_va", "W_ha", "W_a",
                                         "W_x", "W_h", "W_s", "output_W", "output_b"};
    bool known = false;
    if (kind == LRP_DEC_ADAPTIVE) { for (const char* k : adaptive_names) known |= nm == k; }
    else { for (const char* k : gridtd_names) known |= nm == k; }
    if (!known) return fail(LRP_ERR_INVALID, "unknown weight name '%s'", nm.c_str());
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) n *= (size_t)shape[i];
    raw[nm].assign(data, data + n);
    raw_shape[nm].assign(shape, shape + ndim);
    finalized = false;
    return LRP_OK;
  }

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

  int finalize(int64_t* total) {
    if (finalized) return LRP_OK;
    if (kind != LRP_DEC_ADAPTIVE) return fail(LRP_ERR_UNSUPPORTED, "grid-TD decoder not built yet");
    LRP_TRY(need("image_features_W", {D, H})); LRP_TRY(need("image_features_b", {H}));
    LRP_TRY(need("global_W", {D, E})); LRP_TRY(need("global_b", {E}));
    LRP_TRY(need("embedding", {V, E}));
    LRP_TRY(need("lstm_Wi", {2 * E, 4 * H})); LRP_TRY(need("lstm_Wh", {H, 4 * H})); LRP_TRY(need("lstm_b", {4 * H}));
    LRP_TRY(need("Wv", {H, H})); LRP_TRY(need("Wg", {H, H})); LRP_TRY(need("V", {H}));
    LRP_TRY(need("Wx", {2 * E, H})); LRP_TRY(need("Wh", {H, H})); LRP_TRY(need("Ws", {H, H}));
    LRP_TRY(need("output_W", {H, V})); LRP_TRY(need("output_b", {V}));
    const std::vector<float>&Wif = raw["image_features_W"], &Wi = raw["lstm_Wi"], &Wh = raw["lstm_Wh"], &Wx = raw["Wx"],
                            &Whs = raw["Wh"], &Wgl = raw["global_W"];
    std::vector<float> pk;
    {  // image_features as a 1-tap dual conv: cols [0,H) -> relu (V), cols [H,2H) -> pre-activation
      const int Np = conv_npad(2 * H), K = conv_cinp(D);
      pk.assign((size_t)Np * K, 0.f);
      pack_conv_fwd(Wif.data(), 1, D, H, 0, Np, pk.data());
      pack_conv_fwd(Wif.data(), 1, D, H, H, Np, pk.data());
      LRP_TRY(upload(w_if_dual, pk, total));
    }
    {
      const int Np = conv_npad(H), K = conv_cinp(H);
      pk.assign((size_t)Np * K, 0.f);
      pack_conv_fwd(raw["Wv"].data(), 1, H, H, 0, Np, pk.data());
      LRP_TRY(upload(w_v, pk, total));
    }
    LRP_TRY(upload(zero_bias, std::vector<float>(std::max(H, E), 0.f), total));
    LRP_TRY(upload(b_if, raw["image_features_b"], total));
    {  // [x | h_prev] . [[Wi | Wx] ; [Wh | Wh_sentinel]]  -> 4H gate pre-activations + H sentinel gate
      const int Kd = 2 * E + H, N5 = 5 * H;
      pk.assign((size_t)Kd * N5, 0.f);
      for (int k = 0; k < Kd; ++k)
        for (int n = 0; n < N5; ++n) {
          float v;
          if (k < 2 * E) v = n < 4 * H ? Wi[(size_t)k * 4 * H + n] : Wx[(size_t)k * H + n - 4 * H];
          else v = n < 4 * H ? Wh[(size_t)(k - 2 * E) * 4 * H + n] : Whs[(size_t)(k - 2 * E) * H + n - 4 * H];
          pk[(size_t)k * N5 + n] = v;
        }
      LRP_TRY(upload(Wcat, pk, total));
      std::vector<float> bc(N5, 0.f);
      std::copy(raw["lstm_b"].begin(), raw["lstm_b"].end(), bc.begin());
      LRP_TRY(upload(bcat, bc, total));
      // transposed gate-g block for the LRP scan: WgT[j][d] = [Wi;Wh][d][2H+j]   (E:556-558)
      pk.assign((size_t)H * Kd, 0.f);
      for (int d = 0; d < Kd; ++d)
        for (int j = 0; j < H; ++j)
          pk[(size_t)j * Kd + d] = d < 2 * E ? Wi[(size_t)d * 4 * H + 2 * H + j] : Wh[(size_t)(d - 2 * E) * 4 * H + 2 * H + j];
      LRP_TRY(upload(WgT, pk, total));
    }
    LRP_TRY(upload(Wg, raw["Wg"], total));
    LRP_TRY(upload(Ws, raw["Ws"], total));
    LRP_TRY(upload(vvec, raw["V"], total));
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
    finalized = true;
    return LRP_OK;
  }

  template <typename TX, typename TA, typename TY>
  static hipError_t skinny(const TX* X, int ldx, const float* W, int ldw, const float* bias, TY* Y, int ldy, int R, int K,
                           int N, int relu, hipStream_t st) {
    const dim3 grid((N + 63) / 64, (R + 31) / 32);
    hipLaunchKernelGGL((skinny_gemm_kernel<TX, TA, TY>), grid, dim3(256), 0, st, X, ldx, W, ldw, bias, Y, ldy, R, K, N, relu);
    return hipGetLastError();
  }

  template <typename T>
  T* S_(const char* nm) { return state[nm].buf.as<T>(); }

  // _forward_beam_search for B images (E:370-436)
  int forward(const float* feat_dev, const int32_t* caps, const int32_t* lens, int B, hipStream_t st) {
    int64_t dummy = 0;
    LRP_TRY(finalize(&dummy));
    if (B > B_max) return fail(LRP_ERR_INVALID, "B=%d > max_images=%d", B, B_max);
    int Tmax = 0;
    for (int b = 0; b < B; ++b) {
      if (lens[b] < 1 || lens[b] > Tm) return fail(LRP_ERR_INVALID, "caption %d: length %d outside [1,%d]", b, lens[b], Tm);
      for (int i = 0; i < lens[b]; ++i)
        if (caps[b * Tm + i] < 1 || caps[b * Tm + i] > V)
          return fail(LRP_ERR_INVALID, "caption %d: token id %d outside [1,%d]", b, caps[b * Tm + i], V);
      Tmax = std::max(Tmax, (int)lens[b]);
    }
    LRP_HIP_CHECK(hipStreamSynchronize(st));
    for (int b = 0; b < B; ++b) {
      len_host[b] = lens[b];
      for (int i = 0; i < Tm; ++i) cap_host[b * Tm + i] = i < lens[b] ? caps[b * Tm + i] : eos;
    }
    LRP_HIP_CHECK(hipMemcpyAsync(cap_dev.p, cap_host.data(), (size_t)B * Tm * sizeof(int), hipMemcpyHostToDevice, st));
    for (auto& kv : state) LRP_HIP_CHECK(hipMemsetAsync(kv.second.buf.p, 0, kv.second.buf.bytes, st));
    LRP_HIP_CHECK(hipMemsetAsync(u.p, 0, u.bytes, st));

    // ---- static image part (E:375-388)
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
    hipLaunchKernelGGL(mean_rows_kernel, dim3(B), dim3(256), 0, st, feat_dev, avg.as<float>(), L, D);
    LRP_HIP_CHECK(hipGetLastError());
    LRP_HIP_CHECK((skinny<float, float, float>(avg.as<float>(), D, Wglob.as<float>(), E, bglob.as<float>(),
                                               glob_pre.as<float>(), E, B, D, E, 0, st)));
    // ---- step loop (E:399-436)
    const int S = Tm + 1, Kd = 2 * E + H;
    float* ht = S_<float>("ht");
    float* stt = S_<float>("st");
    for (int i = 0; i < Tmax; ++i) {
      hipLaunchKernelGGL(dec_prep_x_kernel, dim3(B), dim3(256), 0, st, emb.as<float>(), glob_pre.as<float>(), ht,
                         cap_dev.as<int>(), xh.as<float>(), S_<float>("xt"), i, Tm, E, H, V, sos);
      LRP_HIP_CHECK(hipGetLastError());
      LRP_HIP_CHECK((skinny<float, float, float>(xh.as<float>(), Kd, Wcat.as<float>(), 5 * H, bcat.as<float>(),
                                                 zgate.as<float>(), 5 * H, B, Kd, 5 * H, 0, st)));
      hipLaunchKernelGGL(dec_pointwise_kernel, dim3(B), dim3(256), 0, st, zgate.as<float>(), ht, S_<float>("ct"),
                         S_<float>("gt"), S_<float>("it_act"), S_<float>("ft_act"), stt, i, Tm, H);
      LRP_HIP_CHECK(hipGetLastError());
      LRP_HIP_CHECK((skinny<float, float, float>(ht + (size_t)(i + 1) * H, S * H, Wg.as<float>(), H, nullptr,
                                                 hproj.as<float>(), H, B, H, H, 0, st)));
      LRP_HIP_CHECK((skinny<float, float, float>(stt + (size_t)(i + 1) * H, S * H, Ws.as<float>(), H, nullptr,
                                                 sproj.as<float>(), H, B, H, H, 0, st)));
      const size_t lds = (size_t)(2 * H + L + 8) * sizeof(float);
      hipLaunchKernelGGL(dec_attention_kernel, dim3(B), dim3(256), lds, st, hproj.as<float>(), sproj.as<float>(),
                         stat.as<float>(), vvec.as<float>(), if_pre.as<float>(), ht, stt, S_<float>("attention"),
                         S_<float>("beta"), S_<double>("context"), S_<double>("c_hat"), u.as<double>(), i, Tm, L, H);
      LRP_HIP_CHECK(hipGetLastError());
    }
    // ---- output layer for every step at once (E:421-422), float64 like the reference
    LRP_HIP_CHECK((skinny<double, double, double>(u.as<double>(), H, Wout.as<float>(), V, bout.as<float>(),
                                                  S_<double>("caption_preds"), V, B * Tm, H, V, 0, st)));
    B_cur = B;
    have_forward = true;
    return LRP_OK;
  }

  int check_token(int b, int t) const {
    if (!have_forward) return fail(LRP_ERR_STATE, "lrp_decoder_forward must run before explain");
    if (b < 0 || b >= B_cur) return fail(LRP_ERR_INVALID, "image index %d outside the %d forwarded captions", b, B_cur);
    if (t < 1 || t > len_host[b]) return fail(LRP_ERR_RANGE, "index out of range of captions (t=%d, caption length %d)", t, len_host[b]);
    return LRP_OK;
  }

  int explain(int n, const int* img_dev, const int* t_dev, const int32_t*, const int32_t*, int variant,
              const float* feat_dev, float* R_feat_dev, float* att_dev, double* rwords_dev, hipStream_t st) {
    if (kind != LRP_DEC_ADAPTIVE) return fail(LRP_ERR_UNSUPPORTED, "grid-TD decoder not built yet");
    if (variant != LRP_EXPLAIN_SEQUENCE && variant != LRP_EXPLAIN_SINGLE_STEP) return fail(LRP_ERR_INVALID, "bad variant");
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
    const size_t lds = (size_t)(2 * H + std::max(H, E) + E + 8) * sizeof(double);
    hipLaunchKernelGGL(dec_explain_adaptive_kernel, dim3(n), dim3(256), lds, st, a);
    LRP_HIP_CHECK(hipGetLastError());
    TailArgs ta{};
    ta.img_idx = img_dev; ta.tpos = t_dev; ta.F = feat_dev; ta.if_pre = if_pre.as<float>(); ta.att = a.att; ta.ctx = a.ctx;
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
    else if (nm == "total_static_img_feature") src = &stat;
    if (!src) return fail(LRP_ERR_INVALID, "unknown state array '%s'", name);
    if (out_bytes > src->bytes) return fail(LRP_ERR_INVALID, "state '%s' holds %zu bytes, %zu requested", name, src->bytes, out_bytes);
    LRP_HIP_CHECK(hipMemcpyAsync(out_dev, src->p, out_bytes, hipMemcpyDeviceToDevice, st));
    return LRP_OK;
  }
};

}  // namespace lrp
