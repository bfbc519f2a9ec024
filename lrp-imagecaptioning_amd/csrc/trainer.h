// Fine-tune step of the LRP-inference training loop (reference: train.py:573-581,
// `keras_model.train_on_batch(X + [lrp_weight], [y, y])` on the model of models/model.py:1340-1374).
//
// What the reference does per batch: predict (forward), LRPInferenceLayerAdaptive.call (the hot path of this library),
// then a second, training-mode forward + backward + Adam through the whole captioner incl. the VGG16 encoder.  Here
// the encoder forward is the one `lrp_encode_images` already ran for the explanation (the encoder has no
// training-mode layers): its ReLU masks / arg-max routes are the cached LRP gates, its activations are kept
// (`Encoder::keep_acts`).  The step itself:
//   decoder forward with the given dropout masks (train_kernels.h) -> two-headed loss -> decoder backward (hand-written
//   BPTT, weight gradients as K = batch-rows products after the scan) -> d features -> encoder backward = the
//   Gradient walk of the explainer (exact fp32 backward-data convs) with one weight-gradient product per layer and tap
//   hooked in (train_gemm.h, gather form) -> flat gradient buffer (caller's; all-reduced by the host over RCCL) -> Adam.
// Parameters live in ONE flat fp32 master buffer in `param order` (conv W, b per layer, then the decoder weights in
// Decoder::adaptive order); gradients and Adam moments use the same offsets.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <functional>
#include <string>
#include <vector>

#include "common.h"
#include "decoder.h"
#include "encoder.h"
#include "train_gemm.h"
#include "train_gemm_bf16.h"
#include "train_kernels.h"

namespace lrp {

struct TrainParam {
  std::string name;
  size_t off = 0, n = 0;
  std::vector<int64_t> shape;
};

struct Trainer {
  bool ready = false, gridtd = false;
  // LRP_TRAIN_FP32: fp32-grade step (weight gradients on the fp32 MFMA; backward-data convs split-bf16 x3, or exact fp32
  // when the handle runs in LRP_PREC_FP32).  LRP_TRAIN_BF16 (BASELINE config 5 "bf16"): the encoder's weight gradients
  // with bf16 operands on v_mfma_f32_32x32x16_bf16 (train_gemm_bf16.h), fp32 accumulation, fp32 master weights / Adam.
  int train_prec = LRP_TRAIN_FP32;
  int Bm = 0, Tm = 0, L = 0, D = 0, H = 0, E = 0, V = 0;
  std::vector<TrainParam> params;
  size_t n_total = 0;
  DevBuf master, mom, vel;
  std::vector<float> host;                     // staging of the initial weights
  float* host_pinned = nullptr;                // pinned mirror of the slices that are re-packed on the host
  ~Trainer() {
    if (host_pinned) (void)hipHostFree(host_pinned);
    if (ev_fwd) (void)hipEventDestroy(ev_fwd);
  }
  float lr = 0.f, clip = 0.f, b1 = 0.9f, b2 = 0.999f, eps = 1e-7f;
  int64_t iter = 0;
  // forward state (time-major rows (t, b))
  DevBuf Vf, favg, glob, proj, Xall, Z, Gt, Cst, Hst, TC, SU, Sst, HW, SW, ALPHA, BETA, CTX, OUTm, logits, part, losses;
  // backward state
  DevBuf Z2, G2, C2, H2, TC2, CHAT, DZ2, dH2, dC2, DCH, dH2tot;   // grid-TD: language LSTM rows, c_hat rows, carries
  DevBuf X4, H4, P4;                           // LSTM-cell dropout: per-gate masked inputs (4, T, B, 2E) / states (4, T, B, H), partials
  DevBuf Gimg;                                 // A1^T dZ of the image layer (64 x cout) before the x+ / x- halves are folded
  DevBuf Esc, dCtx, dBeta;                     // attention scores / d alpha of the current step, per-step scratch
  DevBuf dOUTm, dHtot, dS, dH, dC, DZ, DZS, DHW, dProj, dVf, dVacc, dX, dglob, dfavg, dF, ws, ident;
  size_t ws_floats = 0;
  bool stepped = false;

  const TrainParam* find(const std::string& nm) const {
    for (const TrainParam& p : params)
      if (p.name == nm) return &p;
    return nullptr;
  }
  float* W(const char* nm) const { return master.as<float>() + find(nm)->off; }
  size_t off(const char* nm) const { return find(nm)->off; }

  int begin(Encoder& enc, Decoder& dec, const lrp_config& c, float lr_, float clip_, float b1_, float b2_, float eps_,
            int64_t* total) {
    gridtd = dec.kind == LRP_DEC_GRIDTD;
    if (ready) return fail(LRP_ERR_STATE, "lrp_train_begin was already called on this handle");
    if (c.E != c.H) return fail(LRP_ERR_UNSUPPORTED, "the fine-tune step needs E == H");
    LRP_TRY(enc.check_ready());
    Bm = c.max_images; Tm = c.max_caption_len; L = c.L; D = c.D; H = c.H; E = c.E; V = c.V;
    if (Bm > c.max_tokens) return fail(LRP_ERR_INVALID, "fine-tune step: max_tokens (%d) must be >= max_images (%d)", c.max_tokens, Bm);
    params.clear();
    n_total = 0;
    auto add = [&](const std::string& nm, std::vector<int64_t> shape) {
      TrainParam p;
      p.name = nm; p.shape = shape; p.n = 1;
      for (int64_t s : shape) p.n *= (size_t)s;
      p.off = n_total;
      n_total += (p.n + 3) / 4 * 4;                                 // 16 B aligned slices
      params.push_back(p);
    };
    for (const ConvLayer& Ly : enc.layers) {
      add(Ly.name + "_W", {3, 3, Ly.cin, Ly.cout});
      add(Ly.name + "_b", {Ly.cout});
    }
    add("image_features_W", {D, H}); add("image_features_b", {H});
    add("global_W", {D, E}); add("global_b", {E});
    add("embedding", {V, E});
    if (gridtd) {
      add("td_Wi", {H + 2 * E, 4 * H}); add("td_Wh", {H, 4 * H}); add("td_b", {4 * H});
      add("lang_Wi", {2 * H, 4 * H}); add("lang_Wh", {H, 4 * H}); add("lang_b", {4 * H});
      add("W_va", {H, H}); add("W_ha", {H, H}); add("W_a", {H, 1}); add("W_x", {H + 2 * E, H}); add("W_h", {H, H}); add("W_s", {H, H});
    } else {
      add("lstm_Wi", {2 * E, 4 * H}); add("lstm_Wh", {H, 4 * H}); add("lstm_b", {4 * H});
      add("Wv", {H, H}); add("Wg", {H, H}); add("V", {H, 1}); add("Wx", {2 * E, H}); add("Wh", {H, H}); add("Ws", {H, H});
    }
    add("output_W", {H, V}); add("output_b", {V});
    host.assign(n_total, 0.f);
    // initial master weights: the host copies of lrp_set_weight, or — for weights that arrived through
    // lrp_set_weight_dev — a device-to-device copy after the upload below
    std::vector<std::pair<size_t, const DevBuf*>> from_dev;
    auto take = [&](size_t pi, const std::vector<float>& hostv, const DevBuf* devv, const char* what) -> int {
      if (hostv.size() == params[pi].n) std::copy(hostv.begin(), hostv.end(), host.begin() + params[pi].off);
      else if (devv && devv->p && devv->bytes == params[pi].n * 4) from_dev.push_back({pi, devv});
      else return fail(LRP_ERR_STATE, "weight '%s' is not set (or has the wrong size)", what);
      return LRP_OK;
    };
    size_t pi = 0;
    for (const ConvLayer& Ly : enc.layers) {
      LRP_TRY(take(pi, Ly.raw_w, &Ly.raw_w_dev, params[pi].name.c_str()));
      LRP_TRY(take(pi + 1, Ly.raw_b, &Ly.raw_b_dev, params[pi + 1].name.c_str()));
      pi += 2;
    }
    static const std::vector<float> none;
    for (; pi < params.size(); ++pi) {
      auto it = dec.raw.find(params[pi].name);
      auto jt = dec.raw_dev.find(params[pi].name);
      LRP_TRY(take(pi, it != dec.raw.end() ? it->second : none, jt != dec.raw_dev.end() ? &jt->second : nullptr, params[pi].name.c_str()));
    }
    if (host_pinned) { (void)hipHostFree(host_pinned); host_pinned = nullptr; }
    if (hipHostMalloc(reinterpret_cast<void**>(&host_pinned), n_total * 4) != hipSuccess) return fail(LRP_ERR_NOMEM, "hipHostMalloc failed");
    LRP_TRY(master.alloc(n_total * 4, total)); LRP_TRY(mom.alloc(n_total * 4, total)); LRP_TRY(vel.alloc(n_total * 4, total));
    LRP_HIP_CHECK(hipMemcpy(master.p, host.data(), n_total * 4, hipMemcpyHostToDevice));
    for (auto& fd : from_dev)
      LRP_HIP_CHECK(hipMemcpy(master.as<float>() + params[fd.first].off, fd.second->p, params[fd.first].n * 4, hipMemcpyDeviceToDevice));
    LRP_HIP_CHECK(hipMemset(mom.p, 0, n_total * 4));
    LRP_HIP_CHECK(hipMemset(vel.p, 0, n_total * 4));
    lr = lr_; clip = clip_; b1 = b1_; b2 = b2_; eps = eps_; iter = 0;
    const size_t B = Bm, T = Tm, TB = B * T;
    LRP_TRY(Vf.alloc(B * L * H * 4, total)); LRP_TRY(favg.alloc(B * D * 4, total)); LRP_TRY(glob.alloc(B * E * 4, total));
    LRP_TRY(proj.alloc(B * L * H * 4, total)); LRP_TRY(Xall.alloc(TB * 2 * E * 4, total)); LRP_TRY(Z.alloc(TB * 5 * H * 4, total));
    LRP_TRY(Gt.alloc(TB * 4 * H * 4, total));
    for (DevBuf* d : {&Cst, &Hst, &TC, &SU, &Sst, &HW, &SW, &CTX, &OUTm, &dOUTm, &DZS, &DHW}) LRP_TRY(d->alloc(TB * H * 4, total));
    LRP_TRY(ALPHA.alloc(TB * L * 4, total)); LRP_TRY(BETA.alloc(TB * 4, total));
    LRP_TRY(logits.alloc(TB * V * 4, total)); LRP_TRY(part.alloc(TB * 5 * 4, total)); LRP_TRY(losses.alloc(32, total));
    for (DevBuf* d : {&dHtot, &dS, &dH, &dC, &dVacc, &dCtx}) LRP_TRY(d->alloc(B * H * 4, total));
    LRP_TRY(Esc.alloc(B * L * 4, total)); LRP_TRY(dBeta.alloc(B * 4, total));
    LRP_TRY(DZ.alloc(TB * 5 * H * 4, total)); LRP_TRY(dProj.alloc(B * L * H * 4, total)); LRP_TRY(dVf.alloc(B * L * H * 4, total));
    LRP_TRY(dX.alloc(TB * 2 * E * 4, total)); LRP_TRY(dglob.alloc(B * E * 4, total)); LRP_TRY(dfavg.alloc(B * D * 4, total));
    LRP_TRY(dF.alloc(B * L * D * 4, total));
    if (gridtd) {                                     // second cell (language LSTM) + c_hat rows
      LRP_TRY(Z2.alloc(TB * 4 * H * 4, total)); LRP_TRY(G2.alloc(TB * 4 * H * 4, total)); LRP_TRY(DZ2.alloc(TB * 4 * H * 4, total));
      for (DevBuf* d : {&C2, &H2, &TC2, &CHAT}) LRP_TRY(d->alloc(TB * H * 4, total));
      for (DevBuf* d : {&dH2, &dC2, &DCH, &dH2tot}) LRP_TRY(d->alloc(B * H * 4, total));
    }
    ws_floats = (size_t)16 << 20;                     // K-split partials; at least two slices of the widest layer's nine taps
    for (const ConvLayer& Ly : enc.layers) ws_floats = std::max(ws_floats, (size_t)2 * 9 * Ly.cin * Ly.cout);
    ws_floats = std::max(ws_floats, (size_t)2 * H * V);
    LRP_TRY(ws.alloc(ws_floats * 4, total));
    LRP_TRY(Gimg.alloc((size_t)64 * enc.layers[0].cout * 4, total));
    std::vector<int> id(Bm);
    for (int i = 0; i < Bm; ++i) id[i] = i;
    LRP_TRY(ident.alloc(Bm * sizeof(int), total));
    LRP_HIP_CHECK(hipMemcpy(ident.p, id.data(), Bm * sizeof(int), hipMemcpyHostToDevice));
    LRP_TRY(enc.enable_keep_acts(total));
    ready = true;
    stepped = false;
    return LRP_OK;
  }

  int mm(bool ta, bool tb, int M, int N, long K, const float* A, long lda, const float* Bp, long ldb, float* C, long ldc, bool acc,
         hipStream_t st) {
    SgemmArgs a{};
    a.A = A; a.B = Bp; a.C = C; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    a.transA = ta; a.transB = tb; a.accumulate = acc;
    LRP_HIP_CHECK(sgemm(a, ws.as<float>(), ws_floats, st));
    return LRP_OK;
  }
  // `nb` products of one shape in one launch: product p on A + p * sA, B + p * sB -> C + p * sC (train_gemm.h: strided batch).
  // The per-gate products of an LSTM step under keras' per-gate dropout masks were four launches (+ four reduce passes) each.
  int mm_batch(int nb, bool ta, bool tb, int M, int N, long K, const float* A, long lda, long sA, const float* Bp, long ldb, long sB,
               float* C, long ldc, long sC, bool acc, hipStream_t st) {
    SgemmArgs a{};
    a.A = A; a.B = Bp; a.C = C; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    a.transA = ta; a.transB = tb; a.accumulate = acc;
    a.taps = nb; a.batchA = sA; a.batchB = sB; a.tapC = sC;
    LRP_HIP_CHECK(sgemm(a, ws.as<float>(), ws_floats, st));
    return LRP_OK;
  }
  // two products of one shape whose operands sit anywhere in device memory: a batch of two whose strides are the operands'
  // distances (the attention's h . W_h and s . W_s of a step, forward and backward)
  int mm_pair(bool ta, bool tb, int M, int N, long K, const float* A0, const float* A1, long lda, const float* B0, const float* B1, long ldb,
              float* C0, float* C1, long ldc, bool acc, hipStream_t st) {
    const long sA = A1 - A0, sB = B1 - B0, sC = C1 - C0;
    return mm_batch(2, ta, tb, M, N, K, A0, lda, sA, B0, ldb, sB, C0, ldc, sC, acc, st);
  }
  static unsigned grid_for(size_t n) { return (unsigned)std::min<size_t>((n + 255) / 256, 4096); }

  // One step: gradients of 0.5 CE(y, logits) + 0.5 CE(y, logits * lrp_weight) for the B images last encoded.
  // cap_in (B, T) embedding rows, y_idx (B, T) class index or -1, lrp_weight (B, T, V); masks null = no dropout:
  // m_if (B, L, H), m_glob (B, E), m_out (B, T, H), m_logits (B, T, V; the grid-TD model's Dropout on the logits, M:1303-1304).
  // LSTM-cell dropout (keras `dropout` / `recurrent_dropout`; the language LSTM for grid-TD): m_lin (T, 4, B, 2E),
  // m_lrec (T, 4, B, H), gate order i f c o.  grads_dev: n_total floats (caller's).  losses_dev: 5 floats.
  struct StepIn {
    const float* feat; int B, T; const int* cap_in; const int* y_idx; const float* lrp_weight;
    const float *m_if, *m_glob, *m_out, *m_lin, *m_lrec, *m_logits;
    float* grads; float* losses_dev; hipStream_t st;
  };

  // (Round 2 carried the two decoder scans as stream-captured hipGraphs behind an env switch.  Replays produced wrong
  //  gradients in one configuration (grid-TD, B = 32, caller on the legacy default stream) and the mechanism was never
  //  found; the measured gain was 1.5 %.  The path was removed in round 3 rather than shipped behind a switch: the scans
  //  are plain launches on the caller's stream, and they read the caller's mask tensors directly.)
  const char* nm_proj() const { return gridtd ? "W_va" : "Wv"; }
  const char* nm_hatt() const { return gridtd ? "W_ha" : "Wg"; }
  const char* nm_satt() const { return gridtd ? "W_s" : "Ws"; }
  const char* nm_vatt() const { return gridtd ? "W_a" : "V"; }

  // The training-mode decoder forward needs only the features, the captions and the dropout masks — not lrp_weight.
  // lrp_train_forward may therefore run it early, on another stream, under the explanation that produces lrp_weight
  // (a chain of small launches under an MFMA-bound walk); lrp_train_step then picks it up (same B, T) and starts at the loss.
  bool fwd_valid = false;
  int fwd_B = 0, fwd_T = 0;
  // what that forward read: lrp_train_step back-propagates through THESE masks, so it insists on being handed the same ones
  const int* fwd_cap = nullptr;
  const float *fwd_m_if = nullptr, *fwd_m_glob = nullptr, *fwd_m_out = nullptr, *fwd_m_lin = nullptr, *fwd_m_lrec = nullptr;
  hipEvent_t ev_fwd = nullptr;
  // An early forward is only valid for the features / weights it ran on.  lrp_encode_images, lrp_set_features and
  // lrp_set_weight call this: a following lrp_train_step then runs its own forward (inline) instead of back-propagating
  // through activations of the OLD batch, and — `st` given — the caller's stream first waits for the side-stream forward,
  // which may still be reading the feature buffer the caller is about to overwrite.
  int drop_early_forward(hipStream_t st) {
    if (!fwd_valid) return LRP_OK;
    fwd_valid = false;
    if (ev_fwd) {
      if (st) LRP_HIP_CHECK(hipStreamWaitEvent(st, ev_fwd, 0));
      else LRP_HIP_CHECK(hipEventSynchronize(ev_fwd));
    }
    return LRP_OK;
  }
  int check_step(Encoder& enc, const StepIn& in) {
    if (!ready) return fail(LRP_ERR_STATE, "lrp_train_begin must run first");
    if (in.B < 1 || in.B > Bm || in.T < 2 || in.T > Tm)
      return fail(LRP_ERR_INVALID, "B=%d / T=%d outside [1,%d] / [2,%d]", in.B, in.T, Bm, Tm);
    if (enc.encoded < in.B || enc.features_only) return fail(LRP_ERR_STATE, "lrp_encode_images must run (after lrp_train_begin) on the batch first");
    if (in.m_logits && !gridtd) return fail(LRP_ERR_INVALID, "the adaptive model has no Dropout on the logits");
    for (size_t li = 0; li + 1 < enc.layers.size(); ++li)
      if (!enc.layers[li].pool_after && !enc.layers[li].Akeep.p) return fail(LRP_ERR_STATE, "activations were not kept");
    return LRP_OK;
  }
  int forward(Encoder& enc, const StepIn& in, int64_t* total) {
    LRP_TRY(check_step(enc, in));
    const int B = in.B, T = in.T;
    hipStream_t st = in.st;
    const size_t TB = (size_t)T * B;
    if ((in.m_lin || in.m_lrec) && !X4.p) {              // allocated on first use
      const size_t cap = (size_t)Bm * Tm;
      LRP_TRY(X4.alloc(4 * cap * 2 * E * 4, total)); LRP_TRY(H4.alloc(4 * cap * H * 4, total)); LRP_TRY(P4.alloc(4 * cap * 2 * E * 4, total));
    }
    const float* feat = in.feat;
    float *vf = Vf.as<float>(), *pj = proj.as<float>(), *x = Xall.as<float>();
    // ---------------- forward: the per-image statics (M:1343-1352, get_constants M:602-604)
    LRP_TRY(mm(false, false, B * L, H, D, feat, D, W("image_features_W"), H, vf, H, false, st));
    hipLaunchKernelGGL(tr_bias_relu_mask_kernel, dim3(grid_for((size_t)B * L * H)), dim3(256), 0, st, vf, W("image_features_b"), in.m_if,
                       (size_t)B * L, H);
    hipLaunchKernelGGL(tr_mean_rows_kernel, dim3(B), dim3(256), 0, st, feat, favg.as<float>(), L, D);
    LRP_TRY(mm(false, false, B, E, D, favg.as<float>(), D, W("global_W"), E, glob.as<float>(), E, false, st));
    hipLaunchKernelGGL(tr_bias_relu_mask_kernel, dim3(grid_for((size_t)B * E)), dim3(256), 0, st, glob.as<float>(), W("global_b"), in.m_glob,
                       (size_t)B, E);
    LRP_TRY(mm(false, false, B * L, H, H, vf, H, W(nm_proj()), H, pj, H, false, st));
    // X rows: [emb | glob] (adaptive input_x) or [glob | emb] (the non-recurrent part of the top-down LSTM's input)
    hipLaunchKernelGGL(tr_build_x_kernel, dim3((unsigned)TB), dim3(256), 0, st, W("embedding"), glob.as<float>(), in.cap_in, x, B, T, E,
                       gridtd ? E : 0, gridtd ? 0 : E);
    LRP_TRY(gridtd ? scan_fwd_gridtd(in) : scan_fwd_adaptive(in));
    LRP_HIP_CHECK(hipGetLastError());
    LRP_TRY(mm(false, false, (int)TB, V, H, OUTm.as<float>(), H, W("output_W"), V, logits.as<float>(), V, false, st));   // logits - bias
    if (!ev_fwd) LRP_HIP_CHECK(hipEventCreateWithFlags(&ev_fwd, hipEventDisableTiming));
    LRP_HIP_CHECK(hipEventRecord(ev_fwd, st));
    fwd_valid = true; fwd_B = B; fwd_T = T;
    fwd_cap = in.cap_in; fwd_m_if = in.m_if; fwd_m_glob = in.m_glob; fwd_m_out = in.m_out; fwd_m_lin = in.m_lin; fwd_m_lrec = in.m_lrec;
    return LRP_OK;
  }

  int step(Encoder& enc, const StepIn& in, int64_t* total) {
    LRP_TRY(check_step(enc, in));
    const int B = in.B, T = in.T;
    hipStream_t st = in.st;
    const size_t TB = (size_t)T * B, BH = (size_t)B * H;
    if (fwd_valid && fwd_B == B && fwd_T == T) {           // ran early (lrp_train_forward)
      // the backward scan must see the captions and dropout masks its forward saw (include/lrp_hip.h: lrp_train_forward)
      if (in.cap_in != fwd_cap || in.m_if != fwd_m_if || in.m_glob != fwd_m_glob || in.m_out != fwd_m_out || in.m_lin != fwd_m_lin ||
          in.m_lrec != fwd_m_lrec) {
        // Refuse, and DROP the pending forward: its buffers may be released by the caller after this error, and a retry
        // whose fresh arrays land on the same addresses must not pass the comparison above (it then runs its own forward).
        LRP_TRY(drop_early_forward(st));
        return fail(LRP_ERR_INVALID, "lrp_train_step: cap_in / dropout masks differ from the ones the pending lrp_train_forward ran with "
                                     "(the pending forward was dropped)");
      }
      LRP_HIP_CHECK(hipStreamWaitEvent(st, ev_fwd, 0));
    } else {
      LRP_TRY(forward(enc, in, total));
    }
    fwd_valid = false;
    const float* feat = in.feat;
    float* grads = in.grads;
    float* vf = Vf.as<float>();
    auto g = [&](const char* nm) { return grads + off(nm); };
    // ---------------- two-headed loss, d logits in place
    float* lg = logits.as<float>();
    const float scale = 1.f / (float)((size_t)B * (T - 1));
    hipLaunchKernelGGL(tr_loss_kernel, dim3((unsigned)TB), dim3(256), 0, st, lg, W("output_b"), in.lrp_weight, in.m_logits, in.y_idx,
                       part.as<float>(), B, T, V, scale);
    hipLaunchKernelGGL(tr_loss_final_kernel, dim3(1), dim3(64), 0, st, part.as<float>(), (int)TB, scale, losses.as<float>());
    LRP_HIP_CHECK(hipGetLastError());
    if (in.losses_dev) LRP_HIP_CHECK(hipMemcpyAsync(in.losses_dev, losses.p, 5 * sizeof(float), hipMemcpyDeviceToDevice, st));
    float* wsf = ws.as<float>();
    LRP_TRY(mm(true, false, H, V, (long)TB, OUTm.as<float>(), H, lg, V, g("output_W"), V, false, st));
    LRP_HIP_CHECK(colsum(lg, V, (long)TB, V, g("output_b"), 0, wsf, ws_floats, st));
    LRP_TRY(mm(false, true, (int)TB, H, V, lg, V, W("output_W"), V, dOUTm.as<float>(), H, false, st));
    LRP_HIP_CHECK(hipMemsetAsync(dProj.p, 0, (size_t)B * L * H * 4, st));
    LRP_HIP_CHECK(hipMemsetAsync(dVf.p, 0, (size_t)B * L * H * 4, st));
    LRP_HIP_CHECK(hipMemsetAsync(dVacc.p, 0, BH * 4, st));
    LRP_HIP_CHECK(hipMemsetAsync(dC.p, 0, BH * 4, st));
    // ---------------- reverse scan + the weight gradients of the recurrent part (K = (t, b) rows); leaves dX
    LRP_TRY(gridtd ? scan_bwd_gridtd(in) : scan_bwd_adaptive(in));
    LRP_HIP_CHECK(hipGetLastError());
    // ---------------- attention statics, embedding, global / image_features branches
    float* dx = dX.as<float>();
    LRP_TRY(mm(true, false, H, H, (long)TB, Hst.as<float>(), H, DHW.as<float>(), H, g(nm_hatt()), H, false, st));
    LRP_TRY(mm(true, false, H, H, (long)TB, Sst.as<float>(), H, DZS.as<float>(), H, g(nm_satt()), H, false, st));
    LRP_HIP_CHECK(colsum(dVacc.as<float>(), H, B, H, g(nm_vatt()), 0, wsf, ws_floats, st));
    LRP_TRY(mm(true, false, H, H, (long)B * L, vf, H, dProj.as<float>(), H, g(nm_proj()), H, false, st));
    LRP_TRY(mm(false, true, B * L, H, H, dProj.as<float>(), H, W(nm_proj()), H, dVf.as<float>(), H, true, st));
    LRP_HIP_CHECK(hipMemsetAsync(g("embedding"), 0, (size_t)V * E * 4, st));
    hipLaunchKernelGGL(tr_embedding_bwd_kernel, dim3((unsigned)TB), dim3(256), 0, st, dx, in.cap_in, g("embedding"), B, T, E, gridtd ? E : 0);
    hipLaunchKernelGGL(tr_dglob_kernel, dim3(grid_for((size_t)B * E)), dim3(256), 0, st, dx, dglob.as<float>(), B, T, E, gridtd ? 0 : E);
    hipLaunchKernelGGL(tr_relu_mask_bwd_kernel, dim3(grid_for((size_t)B * E)), dim3(256), 0, st, dglob.as<float>(), glob.as<float>(), in.m_glob,
                       (size_t)B * E);
    LRP_TRY(mm(true, false, D, E, B, favg.as<float>(), D, dglob.as<float>(), E, g("global_W"), E, false, st));
    LRP_HIP_CHECK(colsum(dglob.as<float>(), E, B, E, g("global_b"), 0, wsf, ws_floats, st));
    LRP_TRY(mm(false, true, B, D, E, dglob.as<float>(), E, W("global_W"), E, dfavg.as<float>(), D, false, st));
    hipLaunchKernelGGL(tr_relu_mask_bwd_kernel, dim3(grid_for((size_t)B * L * H)), dim3(256), 0, st, dVf.as<float>(), vf, in.m_if,
                       (size_t)B * L * H);
    LRP_TRY(mm(true, false, D, H, (long)B * L, feat, D, dVf.as<float>(), H, g("image_features_W"), H, false, st));
    LRP_HIP_CHECK(colsum(dVf.as<float>(), H, (long)B * L, H, g("image_features_b"), 0, wsf, ws_floats, st));
    LRP_TRY(mm(false, true, B * L, D, H, dVf.as<float>(), H, W("image_features_W"), H, dF.as<float>(), D, false, st));
    hipLaunchKernelGGL(tr_mean_rows_bwd_kernel, dim3(grid_for((size_t)B * L * D)), dim3(256), 0, st, dF.as<float>(), dfavg.as<float>(), L, D,
                       (size_t)B * L * D);
    LRP_HIP_CHECK(hipGetLastError());
    // ---------------- encoder: Gradient walk + one weight-gradient product per layer
    std::function<int(int, const float*)> hook = [&](int li, const float* dZ) -> int {
      const ConvLayer& Ly = enc.layers[li];
      const long K = (long)B * Ly.H * Ly.W;
      float* gw = grads + params[2 * li].off;
      if (li == 0) {
        // the image layer: one product over the im2col matrix the forward already built (64 columns = 27 taps x
        // channels for x+ and again for x-) instead of nine M = 3 products
        float* G = Gimg.as<float>();                // (64 x cout) scratch
        LRP_TRY(mm(true, false, 64, Ly.cout, K, enc.a1.as<float>(), 64, dZ, Ly.cout, G, Ly.cout, false, st));
        hipLaunchKernelGGL(tr_fold_image_wgrad_kernel, dim3((27 * Ly.cout + 255) / 256), dim3(256), 0, st, G, gw, Ly.cout);
      } else {
        SgemmArgs a{};                              // all nine taps in one launch (blockIdx.z = tap x K slice)
        a.A = enc.layer_input(li); a.lda = Ly.cin; a.B = dZ; a.ldb = Ly.cout; a.C = gw; a.ldc = Ly.cout;
        a.M = Ly.cin; a.N = Ly.cout; a.K = K; a.transA = 1; a.transB = 0;
        a.gather = 1; a.gH = Ly.H; a.gW = Ly.W; a.taps = 9; a.tapC = (long)Ly.cin * Ly.cout;
        if (train_prec == LRP_TRAIN_BF16) LRP_HIP_CHECK(wgrad_bf16(a, wsf, ws_floats, st));
        else LRP_HIP_CHECK(sgemm(a, wsf, ws_floats, st));
      }
      LRP_HIP_CHECK(colsum(dZ, Ly.cout, K, Ly.cout, grads + params[2 * li + 1].off, 0, wsf, ws_floats, st));
      return LRP_OK;
    };
    LRP_TRY(enc.explain(B, ident.as<int>(), dF.as<float>(), nullptr, st, 1, &hook));
    stepped = true;
    return LRP_OK;
  }

  // attention of step t for the rows hrow (h or h1) / srow (sentinel): HW, SW, scores, soft-max, beta, context
  int attention_fwd(const StepIn& in, int t, const float* hs_for_out, const float* mask_out, float* out_rows) {
    const int B = in.B, T = in.T;
    const size_t BH = (size_t)B * H;
    hipStream_t st = in.st;
    LRP_TRY(mm_pair(false, false, B, H, H, Hst.as<float>() + t * BH, Sst.as<float>() + t * BH, H, W(nm_hatt()), W(nm_satt()), H,
                    HW.as<float>() + t * BH, SW.as<float>() + t * BH, H, false, st));
    hipLaunchKernelGGL(tr_att_scores_kernel, dim3(B, (L + 3) / 4), dim3(256), 0, st, proj.as<float>(), HW.as<float>() + t * BH, W(nm_vatt()),
                       Esc.as<float>(), L, H);
    hipLaunchKernelGGL(tr_att_mix_kernel, dim3(B, (H + 63) / 64), dim3(256), (size_t)(L + 8 + 256) * 4, st, Esc.as<float>(), Vf.as<float>(),
                       HW.as<float>() + t * BH, SW.as<float>() + t * BH, W(nm_vatt()), hs_for_out, Sst.as<float>() + t * BH, mask_out,
                       ALPHA.as<float>() + (size_t)t * B * L, BETA.as<float>() + (size_t)t * B, CTX.as<float>() + t * BH, out_rows, L, H, T, t);
    return LRP_OK;
  }
  // its backward: d = gradient at c_hat (+ at h when add_to_h); dHin = what already flows into h; leaves dHtot, dS
  int attention_bwd(const StepIn& in, int t, const float* d_rows, const float* mask_out, const float* dHin, int add_to_h) {
    const int B = in.B, T = in.T;
    const size_t BH = (size_t)B * H;
    hipStream_t st = in.st;
    hipLaunchKernelGGL(tr_att_bwd_head_kernel, dim3(B), dim3(256), 0, st, Sst.as<float>() + t * BH, CTX.as<float>() + t * BH,
                       BETA.as<float>() + (size_t)t * B, d_rows, mask_out, dHin, dHtot.as<float>(), dS.as<float>(), dCtx.as<float>(),
                       dBeta.as<float>(), H, T, t, add_to_h);
    hipLaunchKernelGGL(tr_att_bwd_dalpha_kernel, dim3(B, (L + 3) / 4), dim3(256), 0, st, Vf.as<float>(), dCtx.as<float>(), Esc.as<float>(), L, H);
    hipLaunchKernelGGL(tr_att_bwd_main_kernel, dim3(B, (H + 63) / 64), dim3(256), (size_t)(2 * L + 8 + 512) * 4, st, proj.as<float>(),
                       HW.as<float>() + t * BH, SW.as<float>() + t * BH, W(nm_vatt()), ALPHA.as<float>() + (size_t)t * B * L,
                       BETA.as<float>() + (size_t)t * B, Esc.as<float>(), dBeta.as<float>(), dCtx.as<float>(), DZS.as<float>() + t * BH,
                       DHW.as<float>() + t * BH, dProj.as<float>(), dVf.as<float>(), dVacc.as<float>(), L, H);
    LRP_TRY(mm_pair(false, true, B, H, H, DZS.as<float>() + t * BH, DHW.as<float>() + t * BH, H, W(nm_satt()), W(nm_hatt()), H,
                    dS.as<float>(), dHtot.as<float>(), H, true, st));
    return LRP_OK;
  }

  // ---- adaptive attention (ExternalAttentionRNNWrapperLocalAttentionV3.step, M:573-600)
  int scan_fwd_adaptive(const StepIn& in) {
    const int B = in.B, T = in.T;
    const size_t TB = (size_t)T * B, BH = (size_t)B * H, xs = TB * 2 * E, hs = TB * H;
    hipStream_t st = in.st;
    float *x = Xall.as<float>(), *z = Z.as<float>();
    if (in.m_lin) {
      hipLaunchKernelGGL(tr_gate_masks_kernel, dim3(grid_for(xs)), dim3(256), 0, st, x, in.m_lin, X4.as<float>(), (int)TB, B, 2 * E, 0, xs);
      for (int gt = 0; gt < 4; ++gt)
        LRP_TRY(mm(false, false, (int)TB, H, 2 * E, X4.as<float>() + gt * xs, 2 * E, W("lstm_Wi") + gt * H, 4 * H, z + gt * H, 5 * H, false, st));
    } else {
      LRP_TRY(mm(false, false, (int)TB, 4 * H, 2 * E, x, 2 * E, W("lstm_Wi"), 4 * H, z, 5 * H, false, st));
    }
    LRP_TRY(mm(false, false, (int)TB, H, 2 * E, x, 2 * E, W("Wx"), H, z + 4 * H, 5 * H, false, st));   // sentinel: un-dropped input (M:584)
    for (int t = 0; t < T; ++t) {
      float* zt = z + (size_t)t * B * 5 * H;
      const float* hp = t > 0 ? Hst.as<float>() + (size_t)(t - 1) * BH : nullptr;
      if (t > 0) {
        if (in.m_lrec) {
          float* h4 = H4.as<float>() + (size_t)t * BH;
          hipLaunchKernelGGL(tr_gate_masks_kernel, dim3(grid_for(BH)), dim3(256), 0, st, hp, in.m_lrec, h4, B, B, H, t, hs);
          LRP_TRY(mm_batch(4, false, false, B, H, H, h4, H, (long)hs, W("lstm_Wh"), 4 * H, H, zt, 5 * H, H, true, st));
        } else {
          LRP_TRY(mm(false, false, B, 4 * H, H, hp, H, W("lstm_Wh"), 4 * H, zt, 5 * H, true, st));
        }
        LRP_TRY(mm(false, false, B, H, H, hp, H, W("Wh"), H, zt + 4 * H, 5 * H, true, st));
      }
      hipLaunchKernelGGL(tr_cell_fwd_kernel, dim3(grid_for(BH)), dim3(256), 0, st, zt, W("lstm_b"),
                         t > 0 ? Cst.as<float>() + (size_t)(t - 1) * BH : (const float*)nullptr, Gt.as<float>() + (size_t)t * B * 4 * H,
                         Cst.as<float>() + t * BH, Hst.as<float>() + t * BH, TC.as<float>() + t * BH, SU.as<float>() + t * BH,
                         Sst.as<float>() + t * BH, B, H, 5 * H);
      LRP_TRY(attention_fwd(in, t, Hst.as<float>() + t * BH, in.m_out, OUTm.as<float>() + t * BH));
    }
    return LRP_OK;
  }
  int scan_bwd_adaptive(const StepIn& in) {
    const int B = in.B, T = in.T;
    const size_t TB = (size_t)T * B, BH = (size_t)B * H, xs = TB * 2 * E, hs = TB * H;
    hipStream_t st = in.st;
    float *x = Xall.as<float>(), *dz = DZ.as<float>(), *dx = dX.as<float>(), *wsf = ws.as<float>();
    float* grads = in.grads;
    auto g = [&](const char* nm) { return grads + off(nm); };
    for (int t = T - 1; t >= 0; --t) {
      LRP_TRY(attention_bwd(in, t, dOUTm.as<float>() + t * BH, in.m_out, t == T - 1 ? (const float*)nullptr : dH.as<float>(), 1));
      float* dzt = dz + (size_t)t * B * 5 * H;
      hipLaunchKernelGGL(tr_cell_bwd_kernel, dim3(grid_for(BH)), dim3(256), 0, st, Gt.as<float>() + (size_t)t * B * 4 * H,
                         t > 0 ? Cst.as<float>() + (size_t)(t - 1) * BH : (const float*)nullptr, TC.as<float>() + t * BH,
                         SU.as<float>() + t * BH, dHtot.as<float>(), dS.as<float>(), dC.as<float>(), dzt, B, H, 5 * H);
      if (t > 0) {
        if (in.m_lrec) {
          float* p4 = P4.as<float>();
          LRP_TRY(mm_batch(4, false, true, B, H, H, dzt, 5 * H, H, W("lstm_Wh"), 4 * H, H, p4, H, (long)BH, false, st));
          hipLaunchKernelGGL(tr_gate_masks_bwd_kernel, dim3(grid_for(BH)), dim3(256), 0, st, p4, in.m_lrec, dH.as<float>(), B, B, H, t, BH);
        } else {
          LRP_TRY(mm(false, true, B, H, 4 * H, dzt, 5 * H, W("lstm_Wh"), 4 * H, dH.as<float>(), H, false, st));
        }
        LRP_TRY(mm(false, true, B, H, H, dzt + 4 * H, 5 * H, W("Wh"), H, dH.as<float>(), H, true, st));
      }
    }
    if (in.m_lin) {
      float* p4 = P4.as<float>();
      for (int gt = 0; gt < 4; ++gt) {
        LRP_TRY(mm(false, true, (int)TB, 2 * E, H, dz + gt * H, 5 * H, W("lstm_Wi") + gt * H, 4 * H, p4 + gt * xs, 2 * E, false, st));
        LRP_TRY(mm(true, false, 2 * E, H, (long)TB, X4.as<float>() + gt * xs, 2 * E, dz + gt * H, 5 * H, g("lstm_Wi") + gt * H, 4 * H, false, st));
      }
      hipLaunchKernelGGL(tr_gate_masks_bwd_kernel, dim3(grid_for(xs)), dim3(256), 0, st, p4, in.m_lin, dx, (int)TB, B, 2 * E, 0, xs);
    } else {
      LRP_TRY(mm(false, true, (int)TB, 2 * E, 4 * H, dz, 5 * H, W("lstm_Wi"), 4 * H, dx, 2 * E, false, st));
      LRP_TRY(mm(true, false, 2 * E, 4 * H, (long)TB, x, 2 * E, dz, 5 * H, g("lstm_Wi"), 4 * H, false, st));
    }
    LRP_TRY(mm(false, true, (int)TB, 2 * E, H, dz + 4 * H, 5 * H, W("Wx"), H, dx, 2 * E, true, st));
    LRP_TRY(mm(true, false, 2 * E, H, (long)TB, x, 2 * E, dz + 4 * H, 5 * H, g("Wx"), H, false, st));
    const long Kr = (long)(T - 1) * B;
    if (in.m_lrec) {
      for (int gt = 0; gt < 4; ++gt)                  // rows t = 1 .. T-1 of the masked states and of dz
        LRP_TRY(mm(true, false, H, H, Kr, H4.as<float>() + gt * hs + BH, H, dz + (size_t)B * 5 * H + gt * H, 5 * H,
                   g("lstm_Wh") + gt * H, 4 * H, false, st));
    } else {
      LRP_TRY(mm(true, false, H, 4 * H, Kr, Hst.as<float>(), H, dz + (size_t)B * 5 * H, 5 * H, g("lstm_Wh"), 4 * H, false, st));
    }
    LRP_TRY(mm(true, false, H, H, Kr, Hst.as<float>(), H, dz + (size_t)B * 5 * H + 4 * H, 5 * H, g("Wh"), H, false, st));
    LRP_HIP_CHECK(colsum(dz, 5 * H, (long)TB, 4 * H, g("lstm_b"), 0, wsf, ws_floats, st));
    return LRP_OK;
  }

  // ---- grid-TD (ExternalBottomUpAttentionAdaptive.step, M:784-818): top-down cell on [h2_prev | glob | emb] with the
  // sentinel, attention on h1, language cell (keras LSTMCell, with its dropout) on [c_hat | h1], out = h2 + c_hat.
  // Row stores: Hst / Cst / Gt / TC / SU / Sst / Z / DZ = top-down cell (h1 ...), H2 / C2 / G2 / TC2 / Z2 / DZ2 = language cell.
  int scan_fwd_gridtd(const StepIn& in) {
    const int B = in.B, T = in.T, K1 = H + 2 * E;
    const size_t TB = (size_t)T * B, BH = (size_t)B * H, xs = TB * 2 * H, hs = TB * H;
    hipStream_t st = in.st;
    float *x = Xall.as<float>(), *z = Z.as<float>(), *z2 = Z2.as<float>();
    (void)K1;
    // [glob | emb] part of every step's input at once: rows H.. of td_Wi / W_x
    LRP_TRY(mm(false, false, (int)TB, 4 * H, 2 * E, x, 2 * E, W("td_Wi") + (size_t)H * 4 * H, 4 * H, z, 5 * H, false, st));
    LRP_TRY(mm(false, false, (int)TB, H, 2 * E, x, 2 * E, W("W_x") + (size_t)H * H, H, z + 4 * H, 5 * H, false, st));
    for (int t = 0; t < T; ++t) {
      float* zt = z + (size_t)t * B * 5 * H;
      float* z2t = z2 + (size_t)t * B * 4 * H;
      const float* h1p = t > 0 ? Hst.as<float>() + (size_t)(t - 1) * BH : nullptr;
      const float* h2p = t > 0 ? H2.as<float>() + (size_t)(t - 1) * BH : nullptr;
      if (t > 0) {
        LRP_TRY(mm(false, false, B, 4 * H, H, h2p, H, W("td_Wi"), 4 * H, zt, 5 * H, true, st));
        LRP_TRY(mm(false, false, B, H, H, h2p, H, W("W_x"), H, zt + 4 * H, 5 * H, true, st));
        LRP_TRY(mm(false, false, B, 4 * H, H, h1p, H, W("td_Wh"), 4 * H, zt, 5 * H, true, st));
        LRP_TRY(mm(false, false, B, H, H, h1p, H, W("W_h"), H, zt + 4 * H, 5 * H, true, st));
      }
      hipLaunchKernelGGL(tr_cell_fwd_kernel, dim3(grid_for(BH)), dim3(256), 0, st, zt, W("td_b"),
                         t > 0 ? Cst.as<float>() + (size_t)(t - 1) * BH : (const float*)nullptr, Gt.as<float>() + (size_t)t * B * 4 * H,
                         Cst.as<float>() + t * BH, Hst.as<float>() + t * BH, TC.as<float>() + t * BH, SU.as<float>() + t * BH,
                         Sst.as<float>() + t * BH, B, H, 5 * H);
      float* chat = CHAT.as<float>() + t * BH;
      LRP_TRY(attention_fwd(in, t, nullptr, nullptr, chat));                        // c_hat alone, no mask
      // language LSTM on x2 = [c_hat | h1]
      const float* h1 = Hst.as<float>() + t * BH;
      if (in.m_lin) {
        // masked copies of both halves of x2: X4 rows [g][t][b][2H]; products per gate
        float* x4 = X4.as<float>() + (size_t)t * B * 2 * H;
        hipLaunchKernelGGL(tr_gate_masks2_kernel, dim3(grid_for((size_t)B * 2 * H)), dim3(256), 0, st, chat, h1, in.m_lin, x4, B, H, t, xs);
        LRP_TRY(mm_batch(4, false, false, B, H, 2 * H, x4, 2 * H, (long)xs, W("lang_Wi"), 4 * H, H, z2t, 4 * H, H, false, st));
      } else {
        LRP_TRY(mm(false, false, B, 4 * H, H, chat, H, W("lang_Wi"), 4 * H, z2t, 4 * H, false, st));
        LRP_TRY(mm(false, false, B, 4 * H, H, h1, H, W("lang_Wi") + (size_t)H * 4 * H, 4 * H, z2t, 4 * H, true, st));
      }
      if (t > 0) {
        if (in.m_lrec) {
          float* h4 = H4.as<float>() + (size_t)t * BH;
          hipLaunchKernelGGL(tr_gate_masks_kernel, dim3(grid_for(BH)), dim3(256), 0, st, h2p, in.m_lrec, h4, B, B, H, t, hs);
          LRP_TRY(mm_batch(4, false, false, B, H, H, h4, H, (long)hs, W("lang_Wh"), 4 * H, H, z2t, 4 * H, H, true, st));
        } else {
          LRP_TRY(mm(false, false, B, 4 * H, H, h2p, H, W("lang_Wh"), 4 * H, z2t, 4 * H, true, st));
        }
      }
      hipLaunchKernelGGL(tr_cell_fwd_kernel, dim3(grid_for(BH)), dim3(256), 0, st, z2t, W("lang_b"),
                         t > 0 ? C2.as<float>() + (size_t)(t - 1) * BH : (const float*)nullptr, G2.as<float>() + (size_t)t * B * 4 * H,
                         C2.as<float>() + t * BH, H2.as<float>() + t * BH, TC2.as<float>() + t * BH, (float*)nullptr, (float*)nullptr, B, H,
                         4 * H);
      hipLaunchKernelGGL(tr_out_fwd_kernel, dim3(grid_for(BH)), dim3(256), 0, st, H2.as<float>() + t * BH, chat, in.m_out,
                         OUTm.as<float>() + t * BH, B, H, T, t);
    }
    return LRP_OK;
  }
  int scan_bwd_gridtd(const StepIn& in) {
    const int B = in.B, T = in.T;
    const size_t TB = (size_t)T * B, BH = (size_t)B * H, xs = TB * 2 * H, hs = TB * H;
    hipStream_t st = in.st;
    float *x = Xall.as<float>(), *dz = DZ.as<float>(), *dz2 = DZ2.as<float>(), *dx = dX.as<float>(), *wsf = ws.as<float>();
    float* grads = in.grads;
    auto g = [&](const char* nm) { return grads + off(nm); };
    LRP_HIP_CHECK(hipMemsetAsync(dC2.p, 0, BH * 4, st));
    LRP_HIP_CHECK(hipMemsetAsync(dH.p, 0, BH * 4, st));          // dH = carry into h1, dH2 = carry into h2
    for (int t = T - 1; t >= 0; --t) {
      float* dzt = dz + (size_t)t * B * 5 * H;
      float* dz2t = dz2 + (size_t)t * B * 4 * H;
      hipLaunchKernelGGL(tr_out_bwd_kernel, dim3(grid_for(BH)), dim3(256), 0, st, dOUTm.as<float>() + t * BH, in.m_out,
                         t == T - 1 ? (const float*)nullptr : dH2.as<float>(), DCH.as<float>(), dH2tot.as<float>(), B, H, T, t);
      hipLaunchKernelGGL(tr_cell_bwd_kernel, dim3(grid_for(BH)), dim3(256), 0, st, G2.as<float>() + (size_t)t * B * 4 * H,
                         t > 0 ? C2.as<float>() + (size_t)(t - 1) * BH : (const float*)nullptr, TC2.as<float>() + t * BH,
                         (const float*)nullptr, dH2tot.as<float>(), (const float*)nullptr, dC2.as<float>(), dz2t, B, H, 4 * H);
      // back through x2 = [c_hat | h1] (and the cell's input masks): d c_hat += ..., d h1 (carry dH) += ...
      if (in.m_lin) {
        float* p4 = P4.as<float>();
        LRP_TRY(mm_batch(4, false, true, B, 2 * H, H, dz2t, 4 * H, H, W("lang_Wi"), 4 * H, H, p4, 2 * H, (long)B * 2 * H, false, st));
        hipLaunchKernelGGL(tr_gate_masks2_bwd_kernel, dim3(grid_for((size_t)B * 2 * H)), dim3(256), 0, st, p4, in.m_lin, DCH.as<float>(),
                           dH.as<float>(), B, H, t);
      } else {
        LRP_TRY(mm(false, true, B, H, 4 * H, dz2t, 4 * H, W("lang_Wi"), 4 * H, DCH.as<float>(), H, true, st));
        LRP_TRY(mm(false, true, B, H, 4 * H, dz2t, 4 * H, W("lang_Wi") + (size_t)H * 4 * H, 4 * H, dH.as<float>(), H, true, st));
      }
      if (t > 0) {                                    // carry into h2_{t-1} through the recurrent kernel
        if (in.m_lrec) {
          float* p4 = P4.as<float>();
          LRP_TRY(mm_batch(4, false, true, B, H, H, dz2t, 4 * H, H, W("lang_Wh"), 4 * H, H, p4, H, (long)BH, false, st));
          hipLaunchKernelGGL(tr_gate_masks_bwd_kernel, dim3(grid_for(BH)), dim3(256), 0, st, p4, in.m_lrec, dH2.as<float>(), B, B, H, t, BH);
        } else {
          LRP_TRY(mm(false, true, B, H, 4 * H, dz2t, 4 * H, W("lang_Wh"), 4 * H, dH2.as<float>(), H, false, st));
        }
      }
      LRP_TRY(attention_bwd(in, t, DCH.as<float>(), nullptr, dH.as<float>(), 0));
      hipLaunchKernelGGL(tr_cell_bwd_kernel, dim3(grid_for(BH)), dim3(256), 0, st, Gt.as<float>() + (size_t)t * B * 4 * H,
                         t > 0 ? Cst.as<float>() + (size_t)(t - 1) * BH : (const float*)nullptr, TC.as<float>() + t * BH,
                         SU.as<float>() + t * BH, dHtot.as<float>(), dS.as<float>(), dC.as<float>(), dzt, B, H, 5 * H);
      if (t > 0) {
        LRP_TRY(mm(false, true, B, H, 4 * H, dzt, 5 * H, W("td_Wh"), 4 * H, dH.as<float>(), H, false, st));
        LRP_TRY(mm(false, true, B, H, H, dzt + 4 * H, 5 * H, W("W_h"), H, dH.as<float>(), H, true, st));
        LRP_TRY(mm(false, true, B, H, 4 * H, dzt, 5 * H, W("td_Wi"), 4 * H, dH2.as<float>(), H, true, st));      // rows [0, H) of td_Wi: h2_prev
        LRP_TRY(mm(false, true, B, H, H, dzt + 4 * H, 5 * H, W("W_x"), H, dH2.as<float>(), H, true, st));
      }
    }
    // [glob | emb] part of the top-down input: dX rows; weight gradients, K = (t, b) rows
    LRP_TRY(mm(false, true, (int)TB, 2 * E, 4 * H, dz, 5 * H, W("td_Wi") + (size_t)H * 4 * H, 4 * H, dx, 2 * E, false, st));
    LRP_TRY(mm(false, true, (int)TB, 2 * E, H, dz + 4 * H, 5 * H, W("W_x") + (size_t)H * H, H, dx, 2 * E, true, st));
    LRP_TRY(mm(true, false, 2 * E, 4 * H, (long)TB, x, 2 * E, dz, 5 * H, g("td_Wi") + (size_t)H * 4 * H, 4 * H, false, st));
    LRP_TRY(mm(true, false, 2 * E, H, (long)TB, x, 2 * E, dz + 4 * H, 5 * H, g("W_x") + (size_t)H * H, H, false, st));
    const long Kr = (long)(T - 1) * B;
    const float *dz1 = dz + (size_t)B * 5 * H, *dz21 = dz2 + (size_t)B * 4 * H;       // rows t = 1 .. T-1
    LRP_TRY(mm(true, false, H, 4 * H, Kr, H2.as<float>(), H, dz1, 5 * H, g("td_Wi"), 4 * H, false, st));
    LRP_TRY(mm(true, false, H, H, Kr, H2.as<float>(), H, dz1 + 4 * H, 5 * H, g("W_x"), H, false, st));
    LRP_TRY(mm(true, false, H, 4 * H, Kr, Hst.as<float>(), H, dz1, 5 * H, g("td_Wh"), 4 * H, false, st));
    LRP_TRY(mm(true, false, H, H, Kr, Hst.as<float>(), H, dz1 + 4 * H, 5 * H, g("W_h"), H, false, st));
    LRP_HIP_CHECK(colsum(dz, 5 * H, (long)TB, 4 * H, g("td_b"), 0, wsf, ws_floats, st));
    if (in.m_lin) {
      for (int gt = 0; gt < 4; ++gt)
        LRP_TRY(mm(true, false, 2 * H, H, (long)TB, X4.as<float>() + gt * xs, 2 * H, dz2 + gt * H, 4 * H, g("lang_Wi") + gt * H, 4 * H, false, st));
    } else {
      LRP_TRY(mm(true, false, H, 4 * H, (long)TB, CHAT.as<float>(), H, dz2, 4 * H, g("lang_Wi"), 4 * H, false, st));
      LRP_TRY(mm(true, false, H, 4 * H, (long)TB, Hst.as<float>(), H, dz2, 4 * H, g("lang_Wi") + (size_t)H * 4 * H, 4 * H, false, st));
    }
    if (in.m_lrec) {
      for (int gt = 0; gt < 4; ++gt)
        LRP_TRY(mm(true, false, H, H, Kr, H4.as<float>() + gt * hs + BH, H, dz21 + gt * H, 4 * H, g("lang_Wh") + gt * H, 4 * H, false, st));
    } else {
      LRP_TRY(mm(true, false, H, 4 * H, Kr, H2.as<float>(), H, dz21, 4 * H, g("lang_Wh"), 4 * H, false, st));
    }
    LRP_HIP_CHECK(colsum(dz2, 4 * H, (long)TB, 4 * H, g("lang_b"), 0, wsf, ws_floats, st));
    return LRP_OK;
  }

  // keras Adam(lr, clipvalue) on the master weights, then the engine's operand copies are rebuilt from them
  int apply(Encoder& enc, Decoder& dec, const float* grads, int64_t* total, hipStream_t st) {
    if (!ready) return fail(LRP_ERR_STATE, "lrp_train_begin must run first");
    fwd_valid = false;                             // (a forward of the old weights, if any, is stale)
    ++iter;
    const double lr_t = (double)lr * std::sqrt(1.0 - std::pow((double)b2, (double)iter)) / (1.0 - std::pow((double)b1, (double)iter));
    hipLaunchKernelGGL(tr_adam_kernel, dim3(grid_for(n_total)), dim3(256), 0, st, master.as<float>(), grads, mom.as<float>(),
                       vel.as<float>(), n_total, (float)lr_t, clip, b1, b2, eps);
    LRP_HIP_CHECK(hipGetLastError());
    return sync_engine(enc, dec, total, st);
  }

  // Operand copies are rebuilt on the device: the encoder's by cnn_kernels.h pack_*_dev (image layer included), the
  // decoder's by Decoder::refresh_from_device.  Only a decoder that no forward has finalised yet (refresh returns 1)
  // is handed its weights through the host, once.
  int sync_engine(Encoder& enc, Decoder& dec, int64_t* total, hipStream_t st) {
    if (enc.gates_pending) {                       // the side stream may still read the operand copies we replace
      LRP_HIP_CHECK(hipStreamWaitEvent(st, enc.ev_gates, 0));
      enc.gates_pending = false;
    }
    for (size_t li = 0; li < enc.layers.size(); ++li)
      LRP_TRY(enc.repack_conv_from_device((int)li, master.as<float>() + params[2 * li].off, master.as<float>() + params[2 * li + 1].off,
                                          ws.as<float>(), st));
    const size_t dec0 = params[2 * enc.layers.size()].off;                         // [dec0, n): decoder
    std::function<const float*(const char*)> Wd = [&](const char* nm) -> const float* { return W(nm); };
    const int drc = dec.refresh_from_device(Wd, st);   // 1: the decoder has not built its operand copies yet
    if (drc != LRP_OK && drc != 1) return drc;
    if (drc == 1) {                                // no forward has finalised the decoder yet: hand it the weights once
      LRP_HIP_CHECK(hipMemcpyAsync(host_pinned + dec0, master.as<float>() + dec0, (n_total - dec0) * 4, hipMemcpyDeviceToHost, st));
      LRP_HIP_CHECK(hipStreamSynchronize(st));
      for (size_t pi = 2 * enc.layers.size(); pi < params.size(); ++pi)
        LRP_TRY(dec.set_weight(params[pi].name, host_pinned + params[pi].off, (int)params[pi].shape.size(), params[pi].shape.data(), total));
    }
    enc.encoded = 0;                               // caches belong to the old weights
    return LRP_OK;
  }
};

}  // namespace lrp
