// engine.hip — the C ABI of liblrp_hip.so (include/lrp_hip.h).  Thin glue: argument
// checks, host->device staging of the small index arrays, and dispatch into the
// encoder (CNN half) and decoder (LSTM/attention half) orchestration.
#include <hip/hip_runtime.h>

#include <cstring>
#include <exception>
#include <map>
#include <new>
#include <string>
#include <vector>

#include "common.h"
#include "conv_igemm.h"
#include "conv_sparse.h"
#include "decoder.h"
#include "encoder.h"
#include "resnet_encoder.h"
#include "rules_kernels.h"
#include "score_kernels.h"
#include "trainer.h"

namespace lrp {
std::string& last_error_ref() {
  static thread_local std::string e;
  return e;
}
}  // namespace lrp

using namespace lrp;

struct lrp_handle {
  lrp_config cfg;
  int64_t ws_bytes = 0;
  Encoder enc;             // VGG-style encoder (LRP_ENC_VGG)
  ResNetEncoder rn;        // ResNet bottleneck encoder (LRP_ENC_RESNET)
  bool resnet = false;
  Decoder dec;
  Trainer trainer;         // fine-tune step (lrp_train_*), VGG + adaptive attention
  float* feat() { return resnet ? rn.feat.as<float>() : enc.feat.as<float>(); }
  int& encoded() { return resnet ? rn.encoded : enc.encoded; }
  bool& features_only() { return resnet ? rn.features_only : enc.features_only; }
  DevBuf idx_dev;          // staged (img_idx | t) for the current explain call
  DevBuf rfeat_tmp;        // R_feat when the caller does not want it back
  int* idx_pinned = nullptr;
  hipEvent_t ev_idx = nullptr;   // the last H2D copy out of idx_pinned (guards its reuse without a stream sync)
  ~lrp_handle() {
    if (ev_idx) { (void)hipEventSynchronize(ev_idx); (void)hipEventDestroy(ev_idx); }
    if (idx_pinned) (void)hipHostFree(idx_pinned);
  }
};

static hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }

// ---- the two invariants of the boundary (include/lrp_hip.h:10-20), enforced in ONE place for every entry point:
//  (1) no C++ exception crosses the C ABI: bodies run inside guarded(), which maps std::bad_alloc -> LRP_ERR_NOMEM and
//      anything else -> LRP_ERR_INVALID (std::vector / std::string / std::map in the weight setters and the operator
//      entries can throw; `new lrp_handle` too);
//  (2) a handle's work runs on the handle's device: with_handle() makes cfg.device current for the duration of the call
//      and restores the caller's device on return (lazy allocations — Decoder::finalize, the scan / gradient packs, the
//      trainer's buffers — and every kernel launch would otherwise land on whatever device the calling thread had
//      current: a process that drives several GPUs must not have to remember a hipSetDevice per call).
static int fail_nothrow(int code, const char* msg) noexcept {
  try { last_error_ref() = msg ? msg : ""; } catch (...) {}
  return code;
}
template <class F>
static int guarded(F&& body) noexcept {
  try {
    return body();
  } catch (const std::bad_alloc&) {
    return fail_nothrow(LRP_ERR_NOMEM, "out of host memory inside liblrp_hip");
  } catch (const std::exception& e) {
    return fail_nothrow(LRP_ERR_INVALID, e.what());
  } catch (...) {
    return fail_nothrow(LRP_ERR_INVALID, "unknown C++ exception inside liblrp_hip");
  }
}
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  hipError_t err = hipSuccess;
  explicit DeviceGuard(int dev) {
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != dev) {
      err = hipSetDevice(dev);
      switched = err == hipSuccess;
    }
  }
  ~DeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
};
template <class F>
static int with_handle(const lrp_handle* h, F&& body) noexcept {
  return guarded([&]() -> int {
    if (!h) return fail(LRP_ERR_INVALID, "null handle");
    DeviceGuard g(h->cfg.device);
    if (g.err != hipSuccess) return fail(LRP_ERR_HIP, "cannot make device %d current: %s", h->cfg.device, hipGetErrorString(g.err));
    return body();
  });
}

extern "C" {

int lrp_abi_version(void) { return LRP_ABI_VERSION; }
int lrp_reload_switches(void) {
  return guarded([&]() -> int { lrp::sw().load(); return LRP_OK; });
}

int64_t lrp_launch_count(void) { return (int64_t)lrp::g_launch_count.load(std::memory_order_relaxed); }
const char* lrp_last_error(void) { return last_error_ref().c_str(); }

int lrp_create(const lrp_config* cfg, lrp_handle** out) {
  return guarded([&]() -> int {
    if (!cfg || !out) return fail(LRP_ERR_INVALID, "null argument");
    *out = nullptr;
    if (cfg->abi_version != LRP_ABI_VERSION) return fail(LRP_ERR_INVALID, "abi_version %d != %d", cfg->abi_version, LRP_ABI_VERSION);
    if (cfg->decoder != LRP_DEC_ADAPTIVE && cfg->decoder != LRP_DEC_GRIDTD)
      return fail(LRP_ERR_UNSUPPORTED, "unknown decoder kind %d", cfg->decoder);
    if (cfg->max_images < 1 || cfg->max_tokens < 1 || cfg->max_caption_len < 2)
      return fail(LRP_ERR_INVALID, "capacities must be positive (max_caption_len >= 2)");
    LRP_HIP_CHECK(hipSetDevice(cfg->device));
    lrp_handle* h = new lrp_handle();
    h->cfg = *cfg;
    if (cfg->encoder != LRP_ENC_VGG && cfg->encoder != LRP_ENC_RESNET) {
      delete h;
      return fail(LRP_ERR_UNSUPPORTED, "unknown encoder kind %d", cfg->encoder);
    }
    h->resnet = cfg->encoder == LRP_ENC_RESNET;
    int rc = h->resnet ? h->rn.init(*cfg, &h->ws_bytes) : h->enc.init(*cfg, &h->ws_bytes);
    if (rc == LRP_OK) rc = h->dec.init(*cfg, &h->ws_bytes);
    if (rc == LRP_OK) rc = h->idx_dev.alloc((size_t)cfg->max_tokens * 2 * sizeof(int), &h->ws_bytes);
    if (rc == LRP_OK) rc = h->rfeat_tmp.alloc((size_t)cfg->max_tokens * cfg->L * cfg->D * sizeof(float), &h->ws_bytes);
    if (rc == LRP_OK && hipHostMalloc(reinterpret_cast<void**>(&h->idx_pinned), (size_t)cfg->max_tokens * 2 * sizeof(int)) != hipSuccess)
      rc = fail(LRP_ERR_NOMEM, "hipHostMalloc for index staging failed");
    if (rc != LRP_OK) {
      delete h;
      return rc;
    }
    *out = h;
    return LRP_OK;
  });
}

int lrp_destroy(lrp_handle* h) {
  return guarded([&]() -> int {
    if (!h) return LRP_OK;
    (void)hipSetDevice(h->cfg.device);
    (void)hipDeviceSynchronize();
    delete h;
    return LRP_OK;
  });
}

int64_t lrp_workspace_bytes(const lrp_handle* h) { return h ? h->ws_bytes : 0; }

static int set_weight_host(lrp_handle* h, const char* name, const float* data, int32_t ndim, const int64_t* shape) {
  if (!h || !name || !data || !shape || ndim < 1 || ndim > 4) return fail(LRP_ERR_INVALID, "bad lrp_set_weight arguments");
  LRP_TRY(h->trainer.drop_early_forward(nullptr));   // an lrp_train_forward of the old weights is stale
  const std::string nm(name);
  if (h->resnet) {
    const int rc = h->rn.set_weight(nm, data, ndim, shape, &h->ws_bytes);
    if (rc != 1) return rc;                       // 1 = not an encoder weight
    return h->dec.set_weight(nm, data, ndim, shape, &h->ws_bytes);
  }
  if (nm.size() > 2 && (nm.compare(nm.size() - 2, 2, "_W") == 0 || nm.compare(nm.size() - 2, 2, "_b") == 0)) {
    const int li = h->enc.find_layer(nm.substr(0, nm.size() - 2));
    if (li >= 0) {
      const ConvLayer& L = h->enc.layers[li];
      if (nm.back() == 'W') {
        if (ndim != 4 || shape[0] != 3 || shape[1] != 3 || shape[2] != L.cin || shape[3] != L.cout)
          return fail(LRP_ERR_INVALID, "%s: expected HWIO (3,3,%d,%d)", name, L.cin, L.cout);
        return h->enc.set_conv_weight(li, data, &h->ws_bytes);
      }
      if (ndim != 1 || shape[0] != L.cout) return fail(LRP_ERR_INVALID, "%s: expected (%d,)", name, L.cout);
      return h->enc.set_conv_bias(li, data, &h->ws_bytes);
    }
  }
  return h->dec.set_weight(nm, data, ndim, shape, &h->ws_bytes);
}

int lrp_set_weight(lrp_handle* h, const char* name, const float* data_host, int32_t ndim, const int64_t* shape) {
  return with_handle(h, [&]() -> int {
    return set_weight_host(h, name, data_host, ndim, shape);
  });
}

int lrp_set_weight_dev(lrp_handle* h, const char* name, const float* data_dev, int32_t ndim, const int64_t* shape,
                       void* stream) {
  return with_handle(h, [&]() -> int {
    if (!h || !name || !data_dev || !shape || ndim < 1 || ndim > 4) return fail(LRP_ERR_INVALID, "bad lrp_set_weight_dev arguments");
    LRP_TRY(h->trainer.drop_early_forward(nullptr));
    const std::string nm(name);
    hipStream_t st = S(stream);
    // The multi-GPU start-up path: the bundle arrives in HBM over RCCL/xGMI and stays there — D2D copy of the Keras-layout
    // array, then the device packers (cnn_kernels.h pack_*_dev, Decoder::repack_device) build every operand copy.  No
    // device-to-host copy, no stream synchronisation.
    if (!h->resnet) {
      if (nm.size() > 2 && (nm.compare(nm.size() - 2, 2, "_W") == 0 || nm.compare(nm.size() - 2, 2, "_b") == 0)) {
        const int li = h->enc.find_layer(nm.substr(0, nm.size() - 2));
        if (li >= 0) {
          const ConvLayer& L = h->enc.layers[li];
          if (nm.back() == 'W') {
            if (ndim != 4 || shape[0] != 3 || shape[1] != 3 || shape[2] != L.cin || shape[3] != L.cout)
              return fail(LRP_ERR_INVALID, "%s: expected HWIO (3,3,%d,%d)", name, L.cin, L.cout);
            return h->enc.set_conv_weight_dev(li, data_dev, &h->ws_bytes, st);
          }
          if (ndim != 1 || shape[0] != L.cout) return fail(LRP_ERR_INVALID, "%s: expected (%d,)", name, L.cout);
          return h->enc.set_conv_bias_dev(li, data_dev, &h->ws_bytes, st);
        }
      }
      return h->dec.set_weight_dev(nm, data_dev, ndim, shape, &h->ws_bytes, st);
    }
    if (h->dec.known_weight(nm)) return h->dec.set_weight_dev(nm, data_dev, ndim, shape, &h->ws_bytes, st);
    // ResNet encoder units: packed on the device as well (resnet_encoder.h pack_unit_dev)
    const int rc = h->rn.set_weight_dev(nm, data_dev, ndim, shape, &h->ws_bytes, st);
    if (rc != 1) return rc;
    return fail(LRP_ERR_INVALID, "unknown weight '%s'", name);
  });
}

int lrp_encode_images(lrp_handle* h, const float* images_dev, int32_t B, void* stream) {
  return with_handle(h, [&]() -> int {
    if (!h || !images_dev) return fail(LRP_ERR_INVALID, "null argument");
    LRP_TRY(h->trainer.drop_early_forward(S(stream)));   // it read the features this call overwrites
    LRP_TRY(h->resnet ? h->rn.encode(images_dev, B, S(stream)) : h->enc.encode(images_dev, B, S(stream)));
    return h->dec.on_new_features(B);
  });
}

int lrp_set_features(lrp_handle* h, const float* features_dev, int32_t B, void* stream) {
  return with_handle(h, [&]() -> int {
    if (!h || !features_dev) return fail(LRP_ERR_INVALID, "null argument");
    if (B < 1 || B > h->cfg.max_images) return fail(LRP_ERR_INVALID, "B=%d outside [1,%d]", B, h->cfg.max_images);
    LRP_TRY(h->trainer.drop_early_forward(S(stream)));
    LRP_HIP_CHECK(hipMemcpyAsync(h->feat(), features_dev, (size_t)B * h->cfg.L * h->cfg.D * sizeof(float),
                                 hipMemcpyDeviceToDevice, S(stream)));
    h->encoded() = B;
    h->features_only() = true;
    return h->dec.on_new_features(B);
  });
}

int lrp_get_features(lrp_handle* h, float* features_dev, int32_t B, void* stream) {
  return with_handle(h, [&]() -> int {
    if (!h || !features_dev) return fail(LRP_ERR_INVALID, "null argument");
    if (B < 1 || B > h->encoded()) return fail(LRP_ERR_STATE, "only %d images are cached", h->encoded());
    LRP_HIP_CHECK(hipMemcpyAsync(features_dev, h->feat(), (size_t)B * h->cfg.L * h->cfg.D * sizeof(float),
                                 hipMemcpyDeviceToDevice, S(stream)));
    return LRP_OK;
  });
}

int lrp_decoder_forward(lrp_handle* h, const int32_t* captions_host, const int32_t* lengths_host, int32_t B, void* stream) {
  return with_handle(h, [&]() -> int {
    if (!h || !captions_host || !lengths_host) return fail(LRP_ERR_INVALID, "null argument");
    if (B < 1 || B > h->encoded()) return fail(LRP_ERR_STATE, "B=%d but %d images have features cached", B, h->encoded());
    return h->dec.forward(h->feat(), captions_host, lengths_host, B, S(stream));
  });
}

int lrp_decoder_gen_begin(lrp_handle* h, int32_t B, void* stream) {
  return with_handle(h, [&]() -> int {
    if (!h) return fail(LRP_ERR_INVALID, "null handle");
    if (B < 1 || B > h->encoded()) return fail(LRP_ERR_STATE, "B=%d but %d images have features cached", B, h->encoded());
    return h->dec.gen_begin(h->feat(), B, S(stream));
  });
}

int lrp_decoder_gen_step(lrp_handle* h, int32_t B, const int32_t* parent_host, const int32_t* word_host, int32_t step,
                         double* logits_dev, void* stream) {
  return with_handle(h, [&]() -> int {
    if (!h || !logits_dev || (step > 0 && (!parent_host || !word_host))) return fail(LRP_ERR_INVALID, "null argument");
    return h->dec.gen_step(B, parent_host, word_host, step, logits_dev, S(stream));
  });
}

int lrp_read_state(lrp_handle* h, const char* name, void* out_dev, size_t out_bytes, void* stream) {
  return with_handle(h, [&]() -> int {
    if (!h || !name || !out_dev) return fail(LRP_ERR_INVALID, "null argument");
    return h->dec.read_state(name, out_dev, out_bytes, S(stream));
  });
}

// stage (img_idx | t) into device memory; validates ranges against the cached captions
static int stage_indices(lrp_handle* h, int n, const int32_t* img_idx, const int32_t* t, bool need_t, hipStream_t st) {
  if (n < 1 || n > h->cfg.max_tokens) return fail(LRP_ERR_INVALID, "n=%d outside [1,%d]", n, h->cfg.max_tokens);
  // the pinned staging buffer may still feed the previous call's copy: wait for THAT copy only (no stream sync —
  // the stream usually holds the decoder replay at this point, and draining it would leave the GPU idle while the
  // host enqueues the explain launches)
  if (!h->ev_idx) LRP_HIP_CHECK(hipEventCreateWithFlags(&h->ev_idx, hipEventDisableTiming));
  else LRP_HIP_CHECK(hipEventSynchronize(h->ev_idx));
  for (int i = 0; i < n; ++i) {
    if (img_idx[i] < 0 || img_idx[i] >= h->encoded())
      return fail(LRP_ERR_INVALID, "img_idx[%d]=%d outside the %d cached images", i, img_idx[i], h->encoded());
    h->idx_pinned[i] = img_idx[i];
    if (need_t) {
      LRP_TRY(h->dec.check_token(img_idx[i], t[i]));
      h->idx_pinned[n + i] = t[i];
    }
  }
  LRP_HIP_CHECK(hipMemcpyAsync(h->idx_dev.p, h->idx_pinned, (size_t)n * (need_t ? 2 : 1) * sizeof(int), hipMemcpyHostToDevice, st));
  LRP_HIP_CHECK(hipEventRecord(h->ev_idx, st));
  return LRP_OK;
}

int lrp_decoder_explain(lrp_handle* h, int32_t n, const int32_t* img_idx_host, const int32_t* t_host, int32_t variant,
                        float* R_feat_dev, float* att_dev, double* r_words_dev, void* stream) {
  return with_handle(h, [&]() -> int {
    if (!h || !img_idx_host || !t_host || !R_feat_dev) return fail(LRP_ERR_INVALID, "null argument");
    LRP_TRY(stage_indices(h, n, img_idx_host, t_host, true, S(stream)));
    return h->dec.explain(n, h->idx_dev.as<int>(), h->idx_dev.as<int>() + n, img_idx_host, t_host, variant,
                          h->feat(), R_feat_dev, att_dev, r_words_dev, S(stream));
  });
}

int lrp_cnn_explain(lrp_handle* h, int32_t n, const int32_t* img_idx_host, const float* R_feat_dev, float* R_img_dev,
                    void* stream) {
  return with_handle(h, [&]() -> int {
    if (!h || !img_idx_host || !R_feat_dev || !R_img_dev) return fail(LRP_ERR_INVALID, "null argument");
    if (h->encoded() < 1 || h->features_only()) return fail(LRP_ERR_STATE, "lrp_encode_images must run before lrp_cnn_explain");
    LRP_TRY(stage_indices(h, n, img_idx_host, nullptr, false, S(stream)));
    if (h->resnet) return h->rn.explain(n, h->idx_dev.as<int>(), R_feat_dev, R_img_dev, S(stream));
    h->enc.row2img_host = h->idx_pinned;
    return h->enc.explain(n, h->idx_dev.as<int>(), R_feat_dev, R_img_dev, S(stream));
  });
}

int lrp_explain_tokens(lrp_handle* h, int32_t n, const int32_t* img_idx_host, const int32_t* t_host, int32_t variant,
                       float* R_img_dev, float* R_feat_dev, float* att_dev, double* r_words_dev, void* stream) {
  return with_handle(h, [&]() -> int {
    if (!h || !img_idx_host || !t_host || !R_img_dev) return fail(LRP_ERR_INVALID, "null argument");
    LRP_TRY(stage_indices(h, n, img_idx_host, t_host, true, S(stream)));
    float* rf = R_feat_dev ? R_feat_dev : h->rfeat_tmp.as<float>();
    LRP_TRY(h->dec.explain(n, h->idx_dev.as<int>(), h->idx_dev.as<int>() + n, img_idx_host, t_host, variant,
                           h->feat(), rf, att_dev, r_words_dev, S(stream)));
    if (h->resnet) return h->rn.explain(n, h->idx_dev.as<int>(), rf, R_img_dev, S(stream));
    h->enc.row2img_host = h->idx_pinned;
    return h->enc.explain(n, h->idx_dev.as<int>(), rf, R_img_dev, S(stream));
  });
}

int lrp_decoder_gradient(lrp_handle* h, int32_t n, const int32_t* img_idx_host, const int32_t* t_host, float* d_feat_dev,
                         double* r_words_dev, void* stream) {
  return with_handle(h, [&]() -> int {
    if (!h || !img_idx_host || !t_host || !d_feat_dev) return fail(LRP_ERR_INVALID, "null argument");
    LRP_TRY(stage_indices(h, n, img_idx_host, t_host, true, S(stream)));
    int t_max = 0;
    for (int i = 0; i < n; ++i) t_max = t_host[i] > t_max ? t_host[i] : t_max;
    return h->dec.gradient(n, h->idx_dev.as<int>(), h->idx_dev.as<int>() + n, t_max, d_feat_dev, r_words_dev, &h->ws_bytes,
                           S(stream));
  });
}

int lrp_cnn_walk(lrp_handle* h, int32_t n, const int32_t* img_idx_host, const float* head_dev, float* out_dev, int32_t walk,
                 void* stream) {
  return with_handle(h, [&]() -> int {
    if (!h || !img_idx_host || !head_dev || !out_dev) return fail(LRP_ERR_INVALID, "null argument");
    if (walk < LRP_WALK_LRP || walk > LRP_WALK_GUIDED_BACKPROP) return fail(LRP_ERR_INVALID, "unknown walk %d", walk);
    if (h->encoded() < 1 || h->features_only()) return fail(LRP_ERR_STATE, "lrp_encode_images must run before lrp_cnn_walk");
    if (h->resnet && walk != LRP_WALK_LRP) return fail(LRP_ERR_UNSUPPORTED, "gradient walks exist for the conv-list (VGG) encoder only");
    LRP_TRY(stage_indices(h, n, img_idx_host, nullptr, false, S(stream)));
    if (h->resnet) return h->rn.explain(n, h->idx_dev.as<int>(), head_dev, out_dev, S(stream));
    h->enc.row2img_host = h->idx_pinned;
    return h->enc.explain(n, h->idx_dev.as<int>(), head_dev, out_dev, S(stream), walk);
  });
}

int lrp_set_precision(lrp_handle* h, int32_t mode) {
  return with_handle(h, [&]() -> int {
    if (!h) return fail(LRP_ERR_INVALID, "null handle");
    if (mode != LRP_PREC_FP32 && mode != LRP_PREC_BF16X3 && mode != LRP_PREC_BF16X3_FAST && mode != LRP_PREC_F16X2)
      return fail(LRP_ERR_INVALID, "unknown precision mode %d", mode);
    const int km = mode == LRP_PREC_FP32 ? PREC_FP32 : PREC_BF16X3;
    const bool fast = mode == LRP_PREC_BF16X3_FAST, f16 = mode == LRP_PREC_F16X2;
    // The encode caches belong to the arithmetic they were computed in (the gates' denominators Z+ follow the walk's
    // products; the fp32 mode takes another forward altogether): a mode change drops them, so that an explain call
    // without a new lrp_encode_images returns LRP_ERR_STATE instead of walking with gates of the other arithmetic.
    if (h->enc.prec != km || h->enc.fwd_fast != fast || h->enc.walk_f16 != f16) {
      LRP_TRY(h->trainer.drop_early_forward(nullptr));
      h->enc.encoded = 0;
      h->rn.encoded = 0;
    }
    h->enc.prec = km;
    h->enc.fwd_fast = fast;
    h->enc.walk_f16 = f16;                            // (forward, decoder and the ResNet encoder stay as in LRP_PREC_BF16X3)
    h->rn.prec = km;
    h->dec.prec = km;
    return LRP_OK;
  });
}

int lrp_set_fast_layers(lrp_handle* h, int64_t mask) {
  return with_handle(h, [&]() -> int {
    if (!h) return fail(LRP_ERR_INVALID, "null handle");
    const int nl = (int)h->enc.layers.size();
    if (mask < -1 || (mask >= 0 && nl < 63 && (mask >> nl) != 0)) return fail(LRP_ERR_INVALID, "mask 0x%llx names layers beyond the %d convs", (unsigned long long)mask, nl);
    if (mask >= 0 && (mask & 1)) return fail(LRP_ERR_INVALID, "layer 0 (the image layer) has no two-term form");
    if (h->enc.t2_mask_user != mask && h->enc.walk_f16) {   // the denominators follow the walk: the caches belong to the old choice
      LRP_TRY(h->trainer.drop_early_forward(nullptr));
      h->enc.encoded = 0;
    }
    h->enc.t2_mask_user = mask;
    return LRP_OK;
  });
}

int lrp_profile_enable(lrp_handle* h, int32_t on) {
  return with_handle(h, [&]() -> int {
    if (!h) return fail(LRP_ERR_INVALID, "null handle");
    h->enc.profile = on != 0;
    h->rn.profile = on != 0;
    return LRP_OK;
  });
}

int lrp_profile_query(lrp_handle* h, int64_t* n_launches, double* total_ms, double* total_flop) {
  return with_handle(h, [&]() -> int {
    if (!h) return fail(LRP_ERR_INVALID, "null handle");
    return h->enc.profile_query(n_launches, total_ms, total_flop);
  });
}

int lrp_profile_records(lrp_handle* h, int32_t cap, double* ms_out, double* flop_out, int32_t* n_out) {
  return with_handle(h, [&]() -> int {
    if (!h || !ms_out || !flop_out || !n_out || cap < 0) return fail(LRP_ERR_INVALID, "bad lrp_profile_records arguments");
    return h->resnet ? h->rn.profile_records(cap, ms_out, flop_out, n_out) : h->enc.profile_records(cap, ms_out, flop_out, n_out);
  });
}

int lrp_op_conv(const float* in_dev, const float* w_hwio_host, const float* bias_host, const float* aux_dev, float* out_dev,
                int32_t NB, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t taps, int32_t mode, void* stream) {
  return guarded([&]() -> int {
    if (!in_dev || !w_hwio_host || !out_dev) return fail(LRP_ERR_INVALID, "null argument");
    if (taps != 1 && taps != 9) return fail(LRP_ERR_INVALID, "taps must be 1 or 9");
    if (Cin % 4 != 0) return fail(LRP_ERR_UNSUPPORTED, "Cin must be a multiple of 4");
    const bool split = (mode & LRP_CONV_SPLIT_BF16) != 0;   // same op on the split-bf16 MFMA path (fp32 in, fp32 out)
    mode &= ~LRP_CONV_SPLIT_BF16;
    if (mode < 0 || mode > 3) return fail(LRP_ERR_INVALID, "mode must be 0..3");
    if (split && mode == 0) return fail(LRP_ERR_UNSUPPORTED, "the split-bf16 path has no relu epilogue (modes 1..3)");
    const bool bwd = mode >= 2;
    // forward: in has Cin channels, out Cout.  backward: in has Cout channels (S), out Cin (relevance of the input)
    const int inC = bwd ? Cout : Cin, outC = bwd ? Cin : Cout;
    if (inC % 4 != 0) return fail(LRP_ERR_UNSUPPORTED, "input channels must be a multiple of 4");
    const int Np = conv_npad(outC), K = taps * conv_cinp(inC);
    std::vector<float> pk((size_t)Np * K, 0.f);
    if (bwd) pack_conv_bwd(w_hwio_host, taps, Cin, Cout, 0, pk.data());
    else pack_conv_fwd(w_hwio_host, taps, Cin, Cout, 0, Np, pk.data());
    DevBuf wdev, bdev, insplit, wfrag;
    if (split) {
      if ((inC & 7) || (bwd && (outC & 7))) return fail(LRP_ERR_UNSUPPORTED, "split-bf16 path: channels must be multiples of 8");
      std::vector<float> sp(pk.size());
      pack_split8(pk.data(), pk.size(), sp.data());
      pk.swap(sp);
      if (bwd && taps == 9 && Np == 64) {                  // weights-in-registers variant of the N <= 64 backward convs
        std::vector<float> fr((size_t)64 * K);
        pack_frag64(pk.data(), 9, conv_cinp(inC), fr.data());
        LRP_TRY(wfrag.alloc(fr.size() * sizeof(float), nullptr));
        LRP_HIP_CHECK(hipMemcpy(wfrag.p, fr.data(), fr.size() * sizeof(float), hipMemcpyHostToDevice));
      }
      const size_t n8 = (size_t)NB * H * W * inC / 8;
      LRP_TRY(insplit.alloc(n8 * 32, nullptr));
      hipLaunchKernelGGL(split_copy_kernel, dim3(stream_grid(n8)), dim3(256), 0, S(stream), in_dev, insplit.as<float>(), n8);
      LRP_HIP_CHECK(hipGetLastError());
      in_dev = insplit.as<float>();
    }
    LRP_TRY(wdev.alloc(pk.size() * sizeof(float), nullptr));
    LRP_HIP_CHECK(hipMemcpy(wdev.p, pk.data(), pk.size() * sizeof(float), hipMemcpyHostToDevice));
    ConvArgs ca{};
    ca.out_plain = 1;
    ca.wpk_frag = wfrag.as<float>();
    ca.in = in_dev; ca.wpk = wdev.as<float>(); ca.NB = NB; ca.H = H; ca.W = W; ca.Cin = inC; ca.CinP = conv_cinp(inC);
    ca.N = outC; ca.taps = taps; ca.out = out_dev; ca.aux = aux_dev;
    if (!bwd) {
      if (!bias_host) return fail(LRP_ERR_INVALID, "bias required for forward modes");
      LRP_TRY(bdev.alloc((size_t)Cout * sizeof(float), nullptr));
      LRP_HIP_CHECK(hipMemcpy(bdev.p, bias_host, (size_t)Cout * sizeof(float), hipMemcpyHostToDevice));
      ca.bias = bdev.as<float>();
    } else if (!aux_dev) {
      return fail(LRP_ERR_INVALID, "aux (gate) required for backward modes");
    }
    static const int epi_of_mode[4] = {EPI_BIAS_RELU, EPI_BIAS, EPI_MUL, EPI_MUL_UP2};
    LRP_HIP_CHECK(conv_launch(epi_of_mode[mode], ca, S(stream), split ? PREC_BF16X3 : PREC_FP32));
    LRP_HIP_CHECK(hipStreamSynchronize(S(stream)));      // weights are freed on return
    return LRP_OK;
  });
}

int lrp_op_conv_pool_sparse(const float* sc_dev, const unsigned char* pos_dev, const float* w_hwio_host, const float* gate_dev,
                            float* out_dev, int32_t NB, int32_t Hp, int32_t Wp, int32_t Cin, int32_t Cout, int32_t reps, void* stream) {
  return guarded([&]() -> int {
    if (!sc_dev || !pos_dev || !w_hwio_host || !gate_dev || !out_dev) return fail(LRP_ERR_INVALID, "null argument");
    if (NB < 1 || Hp < 1 || Wp < 1 || (reps & 255) < 1) return fail(LRP_ERR_INVALID, "NB, Hp, Wp, reps must be >= 1");
    if (!conv_sparse_supports(Cin, Cout, Hp, Wp))
      return fail(LRP_ERR_UNSUPPORTED, "the sparse consumer needs Cin %% 256 == 0 (output columns) and Cout %% 16 == 0");
    hipStream_t st = S(stream);
    const int Npb = conv_npad(Cin), Kb = 9 * conv_cinp(Cout);
    std::vector<float> pk((size_t)Npb * Kb, 0.f);
    pack_conv_bwd(w_hwio_host, 9, Cin, Cout, 0, pk.data());
    DevBuf wb, wsp, pairs;
    LRP_TRY(wb.alloc(pk.size() * sizeof(float), nullptr));
    LRP_HIP_CHECK(hipMemcpy(wb.p, pk.data(), pk.size() * sizeof(float), hipMemcpyHostToDevice));
    LRP_TRY(wsp.alloc(conv_sparse_weight_floats(Cin, Cout) * sizeof(float), nullptr));
    LRP_HIP_CHECK(conv_sparse_pack(wb.as<float>(), wsp.as<float>(), Cin, Cout, st));
    const size_t n8 = (size_t)NB * Hp * Wp * Cout / 8;
    LRP_TRY(pairs.alloc(n8 * 32, nullptr));
    LRP_HIP_CHECK(conv_sparse_pairs(sc_dev, pairs.as<float>(), NB, Hp, Wp, Cout, st));
    DevBuf idxp;
    LRP_TRY(idxp.alloc(conv_sparse_index_words(NB, Hp, Wp, Cout) * sizeof(unsigned), nullptr));
    LRP_HIP_CHECK(conv_sparse_index(pos_dev, idxp.as<unsigned>(), NB, Hp, Wp, Cout, st));
    SparseArgs sa{};
    sa.sc = pairs.as<float>(); sa.idxp = idxp.as<unsigned>(); sa.wsp = wsp.as<float>(); sa.gate = gate_dev; sa.out = out_dev;
    sa.NB = NB; sa.Hp = Hp; sa.Wp = Wp; sa.C = Cout; sa.N = Cin; sa.out_plain = 1;
    sa.diag = reps >> 8;                                   // (measurement variants, profiles/sparse_ab.py; results are then not meaningful)
    reps &= 255;
    for (int r = 0; r < reps; ++r) LRP_HIP_CHECK(conv_sparse_launch(sa, st));
    LRP_HIP_CHECK(hipStreamSynchronize(st));               // the operand copies are freed on return
    return LRP_OK;
  });
}

int lrp_op_epsilon_dense(const float* x_dev, const float* W_host, const float* R_dev, float* out_dev, int32_t N,
                         int32_t Din, int32_t Dout, float epsilon, void* stream) {
  return guarded([&]() -> int {
    if (!x_dev || !W_host || !R_dev || !out_dev) return fail(LRP_ERR_INVALID, "null argument");
    if (N < 1 || (Din & 3) || (Dout & 3) || !(epsilon > 0.f)) return fail(LRP_ERR_INVALID, "need N>=1, Din%%4==0, Dout%%4==0, epsilon>0");
    hipStream_t st = S(stream);
    // Z = x.W  (1-tap GEMM, B operand packed [Dout][Din]);  C = S.W^T  (packed [Din][Dout])
    const int Np = conv_npad(Dout), K = conv_cinp(Din), Npb = conv_npad(Din), Kb = conv_cinp(Dout);
    std::vector<float> pk((size_t)Np * K, 0.f), pkb((size_t)Npb * Kb, 0.f);
    pack_conv_fwd(W_host, 1, Din, Dout, 0, Np, pk.data());
    pack_conv_bwd(W_host, 1, Din, Dout, 0, pkb.data());
    DevBuf wf, wb, Z, Sb;
    LRP_TRY(wf.alloc(pk.size() * sizeof(float), nullptr));
    LRP_TRY(wb.alloc(pkb.size() * sizeof(float), nullptr));
    LRP_TRY(Z.alloc((size_t)N * Dout * sizeof(float), nullptr));
    LRP_TRY(Sb.alloc((size_t)N * Dout * sizeof(float), nullptr));
    LRP_HIP_CHECK(hipMemcpy(wf.p, pk.data(), pk.size() * sizeof(float), hipMemcpyHostToDevice));
    LRP_HIP_CHECK(hipMemcpy(wb.p, pkb.data(), pkb.size() * sizeof(float), hipMemcpyHostToDevice));
    ConvArgs cz{};
    cz.in = x_dev; cz.wpk = wf.as<float>(); cz.NB = N; cz.H = 1; cz.W = 1; cz.Cin = Din; cz.CinP = K; cz.N = Dout; cz.taps = 1;
    cz.out = Z.as<float>();
    LRP_HIP_CHECK(conv_launch(EPI_STORE, cz, st));
    const size_t ne = (size_t)N * Dout;
    hipLaunchKernelGGL(eps_divide_kernel, dim3(stream_grid(ne)), dim3(256), 0, st, R_dev, Z.as<float>(), Sb.as<float>(), epsilon, ne);
    LRP_HIP_CHECK(hipGetLastError());
    ConvArgs cb{};
    cb.in = Sb.as<float>(); cb.wpk = wb.as<float>(); cb.NB = N; cb.H = 1; cb.W = 1; cb.Cin = Dout; cb.CinP = Kb; cb.N = Din;
    cb.taps = 1; cb.aux = x_dev; cb.out = out_dev;
    LRP_HIP_CHECK(conv_launch(EPI_MUL, cb, st));                       // R_in = x * (S . W^T)
    LRP_HIP_CHECK(hipStreamSynchronize(st));
    return LRP_OK;
  });
}

int lrp_op_batchnorm_lrp(const float* x_dev, const float* gamma_dev, const float* beta_dev, const float* mean_dev,
                         const float* var_dev, float bn_eps, const float* R_dev, float* out_dev, int64_t n, int32_t C,
                         void* stream) {
  return guarded([&]() -> int {
    if (!x_dev || !gamma_dev || !beta_dev || !mean_dev || !var_dev || !R_dev || !out_dev) return fail(LRP_ERR_INVALID, "null argument");
    if (n < 1 || C < 1 || n % C) return fail(LRP_ERR_INVALID, "n must be a positive multiple of C");
    hipLaunchKernelGGL(bn_lrp_kernel, dim3(stream_grid((size_t)n)), dim3(256), 0, S(stream), x_dev, gamma_dev, beta_dev,
                       mean_dev, var_dev, bn_eps, R_dev, out_dev, (size_t)n, C);
    LRP_HIP_CHECK(hipGetLastError());
    return LRP_OK;
  });
}

int lrp_op_add_lrp(const float* a_dev, const float* b_dev, const float* R_dev, float* Ra_dev, float* Rb_dev, int64_t n,
                   void* stream) {
  return guarded([&]() -> int {
    if (!a_dev || !b_dev || !R_dev || !Ra_dev || !Rb_dev || n < 1) return fail(LRP_ERR_INVALID, "bad argument");
    hipLaunchKernelGGL(add_lrp_kernel, dim3(stream_grid((size_t)n)), dim3(256), 0, S(stream), a_dev, b_dev, R_dev, Ra_dev,
                       Rb_dev, (size_t)n);
    LRP_HIP_CHECK(hipGetLastError());
    return LRP_OK;
  });
}

int lrp_op_avgpool_lrp(const float* x_dev, const float* R_dev, float* out_dev, int32_t NB, int32_t H, int32_t W, int32_t C,
                       int32_t k, void* stream) {
  return guarded([&]() -> int {
    if (!x_dev || !R_dev || !out_dev) return fail(LRP_ERR_INVALID, "null argument");
    if (NB < 1 || k < 1 || H < k || W < k || (H % k) || (W % k)) return fail(LRP_ERR_UNSUPPORTED, "need H, W multiples of the pool size");
    if (C < 4 || (C & 3)) return fail(LRP_ERR_UNSUPPORTED, "C must be a multiple of 4");
    const size_t total = (size_t)NB * (H / k) * (W / k) * (C / 4);
    hipLaunchKernelGGL(avgpool_lrp_kernel, dim3(stream_grid(total)), dim3(256), 0, S(stream), x_dev, R_dev, out_dev, NB, H, W, C, k);
    LRP_HIP_CHECK(hipGetLastError());
    return LRP_OK;
  });
}

int lrp_heatmap_render(const float* R_img_dev, const float* lut_dev, float* rgb_dev, int32_t n, int32_t npix, int32_t C,
                       float gamma, void* stream) {
  return guarded([&]() -> int {
    if (!R_img_dev || !lut_dev || !rgb_dev) return fail(LRP_ERR_INVALID, "null argument");
    if (n < 1 || npix < 1 || C < 1 || !(gamma > 0.f)) return fail(LRP_ERR_INVALID, "n, npix, C, gamma must be positive");
    hipLaunchKernelGGL(heatmap_render_kernel, dim3(n), dim3(256), 0, S(stream), R_img_dev, lut_dev, rgb_dev, npix, C, gamma);
    LRP_HIP_CHECK(hipGetLastError());
    return LRP_OK;
  });
}

int lrp_preprocess_images(const uint8_t* rgb_dev, float* out_dev, int32_t NB, int32_t H0, int32_t W0, int32_t H, int32_t W,
                          void* stream) {
  return guarded([&]() -> int {
    if (!rgb_dev || !out_dev) return fail(LRP_ERR_INVALID, "null argument");
    if (NB < 1 || H0 < 1 || W0 < 1 || H < 1 || W < 1) return fail(LRP_ERR_INVALID, "sizes must be positive");
    hipLaunchKernelGGL(preprocess_caffe_kernel, dim3(stream_grid((size_t)NB * H * W)), dim3(256), 0, S(stream), rgb_dev, out_dev,
                       NB, H0, W0, H, W);
    LRP_HIP_CHECK(hipGetLastError());
    return LRP_OK;
  });
}

// ---- operator-level entries of the fine-tune step's products (unit tests of train_gemm.h at real layer sizes)
int lrp_op_sgemm(const float* A_dev, const float* B_dev, float* C_dev, int32_t M, int32_t N, int64_t K, int64_t lda, int64_t ldb,
                 int64_t ldc, int32_t transA, int32_t transB, int32_t accumulate, float* ws_dev, int64_t ws_floats, void* stream) {
  return guarded([&]() -> int {
    if (!A_dev || !B_dev || !C_dev) return fail(LRP_ERR_INVALID, "null argument");
    if (M < 1 || N < 1 || K < 1 || (transA && transB)) return fail(LRP_ERR_INVALID, "bad sgemm shape / transposes");
    SgemmArgs a{};
    a.A = A_dev; a.B = B_dev; a.C = C_dev; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    a.transA = transA != 0; a.transB = transB != 0; a.accumulate = accumulate != 0;
    LRP_HIP_CHECK(sgemm(a, ws_dev, ws_dev ? (size_t)ws_floats : 0, S(stream)));
    return LRP_OK;
  });
}

static int op_conv_wgrad(bool bf16, const float* x_dev, const float* dz_dev, float* dw_hwio_dev, float* db_dev, int32_t NB, int32_t H,
                         int32_t W, int32_t Cin, int32_t Cout, float* ws_dev, int64_t ws_floats, void* stream) {
  if (!x_dev || !dz_dev || !dw_hwio_dev || !ws_dev) return fail(LRP_ERR_INVALID, "null argument");
  if (NB < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1) return fail(LRP_ERR_INVALID, "sizes must be positive");
  if (ws_floats < (int64_t)2 * 9 * Cin * Cout) return fail(LRP_ERR_INVALID, "workspace smaller than 2 x 9 x Cin x Cout floats");
  SgemmArgs a{};
  a.A = x_dev; a.lda = Cin; a.B = dz_dev; a.ldb = Cout; a.C = dw_hwio_dev; a.ldc = Cout;
  a.M = Cin; a.N = Cout; a.K = (long)NB * H * W; a.transA = 1;
  a.gather = 1; a.gH = H; a.gW = W; a.taps = 9; a.tapC = (long)Cin * Cout;
  if (bf16) LRP_HIP_CHECK(wgrad_bf16(a, ws_dev, (size_t)ws_floats, S(stream)));
  else LRP_HIP_CHECK(sgemm(a, ws_dev, (size_t)ws_floats, S(stream)));
  if (db_dev) LRP_HIP_CHECK(colsum(dz_dev, Cout, (long)NB * H * W, Cout, db_dev, 0, ws_dev, (size_t)ws_floats, S(stream)));
  return LRP_OK;
}

int lrp_op_conv_wgrad(const float* x_dev, const float* dz_dev, float* dw_hwio_dev, float* db_dev, int32_t NB, int32_t H, int32_t W,
                      int32_t Cin, int32_t Cout, float* ws_dev, int64_t ws_floats, void* stream) {
  return guarded([&]() -> int {
    return op_conv_wgrad(false, x_dev, dz_dev, dw_hwio_dev, db_dev, NB, H, W, Cin, Cout, ws_dev, ws_floats, stream);
  });
}

int lrp_op_conv_wgrad_bf16(const float* x_dev, const float* dz_dev, float* dw_hwio_dev, float* db_dev, int32_t NB, int32_t H,
                           int32_t W, int32_t Cin, int32_t Cout, float* ws_dev, int64_t ws_floats, void* stream) {
  return guarded([&]() -> int {
    return op_conv_wgrad(true, x_dev, dz_dev, dw_hwio_dev, db_dev, NB, H, W, Cin, Cout, ws_dev, ws_floats, stream);
  });
}

// ---- fine-tune step, trainer.h
int lrp_train_begin(lrp_handle* h, float lr, float clipvalue, float beta1, float beta2, float eps) {
  return with_handle(h, [&]() -> int {
    if (!h) return fail(LRP_ERR_INVALID, "null handle");
    if (h->resnet) return fail(LRP_ERR_UNSUPPORTED, "the fine-tune step is built for the VGG encoder");
    if (!(lr > 0.f) || clipvalue < 0.f || !(eps > 0.f)) return fail(LRP_ERR_INVALID, "lr, eps must be positive, clipvalue >= 0");
    LRP_HIP_CHECK(hipSetDevice(h->cfg.device));
    return h->trainer.begin(h->enc, h->dec, h->cfg, lr, clipvalue, beta1, beta2, eps, &h->ws_bytes);
  });
}

int lrp_train_set_precision(lrp_handle* h, int32_t mode) {
  return with_handle(h, [&]() -> int {
    if (mode != LRP_TRAIN_FP32 && mode != LRP_TRAIN_BF16) return fail(LRP_ERR_INVALID, "unknown training precision %d", mode);
    h->trainer.train_prec = mode;
    return LRP_OK;
  });
}

int64_t lrp_train_flat_size(const lrp_handle* h) { return h && h->trainer.ready ? (int64_t)h->trainer.n_total : 0; }
int32_t lrp_train_num_params(const lrp_handle* h) { return h && h->trainer.ready ? (int32_t)h->trainer.params.size() : 0; }

int lrp_train_param_info(const lrp_handle* h, int32_t i, const char** name, int64_t* offset, int64_t* size) {
  return with_handle(h, [&]() -> int {
    if (!h || !h->trainer.ready) return fail(LRP_ERR_STATE, "lrp_train_begin must run first");
    if (i < 0 || i >= (int32_t)h->trainer.params.size()) return fail(LRP_ERR_INVALID, "parameter index %d out of range", i);
    const TrainParam& p = h->trainer.params[i];
    if (name) *name = p.name.c_str();
    if (offset) *offset = (int64_t)p.off;
    if (size) *size = (int64_t)p.n;
    return LRP_OK;
  });
}

int lrp_train_step(lrp_handle* h, int32_t B, int32_t T, const int32_t* cap_in_dev, const int32_t* y_idx_dev,
                   const float* lrp_weight_dev, const float* mask_image_features_dev, const float* mask_global_dev,
                   const float* mask_output_dev, const float* mask_lstm_in_dev, const float* mask_lstm_rec_dev,
                   const float* mask_logits_dev, float* grads_dev, float* losses_dev, void* stream) {
  return with_handle(h, [&]() -> int {
    if (!h || !cap_in_dev || !y_idx_dev || !lrp_weight_dev || !grads_dev) return fail(LRP_ERR_INVALID, "null argument");
    Trainer::StepIn in{};
    in.feat = h->feat(); in.B = B; in.T = T; in.cap_in = cap_in_dev; in.y_idx = y_idx_dev; in.lrp_weight = lrp_weight_dev;
    in.m_if = mask_image_features_dev; in.m_glob = mask_global_dev; in.m_out = mask_output_dev; in.m_lin = mask_lstm_in_dev;
    in.m_lrec = mask_lstm_rec_dev; in.m_logits = mask_logits_dev; in.grads = grads_dev; in.losses_dev = losses_dev; in.st = S(stream);
    return h->trainer.step(h->enc, in, &h->ws_bytes);
  });
}

int lrp_train_forward(lrp_handle* h, int32_t B, int32_t T, const int32_t* cap_in_dev, const float* mask_image_features_dev,
                      const float* mask_global_dev, const float* mask_output_dev, const float* mask_lstm_in_dev,
                      const float* mask_lstm_rec_dev, void* stream) {
  return with_handle(h, [&]() -> int {
    if (!h || !cap_in_dev) return fail(LRP_ERR_INVALID, "null argument");
    Trainer::StepIn in{};
    in.feat = h->feat(); in.B = B; in.T = T; in.cap_in = cap_in_dev;
    in.m_if = mask_image_features_dev; in.m_glob = mask_global_dev; in.m_out = mask_output_dev; in.m_lin = mask_lstm_in_dev;
    in.m_lrec = mask_lstm_rec_dev; in.st = S(stream);
    return h->trainer.forward(h->enc, in, &h->ws_bytes);
  });
}

int lrp_train_drop_forward(lrp_handle* h, void* stream) {
  return with_handle(h, [&]() -> int {
    if (!h) return fail(LRP_ERR_INVALID, "null argument");
    return h->trainer.drop_early_forward(S(stream));
  });
}

int lrp_train_apply(lrp_handle* h, const float* grads_dev, void* stream) {
  return with_handle(h, [&]() -> int {
    if (!h || !grads_dev) return fail(LRP_ERR_INVALID, "null argument");
    return h->trainer.apply(h->enc, h->dec, grads_dev, nullptr, S(stream));
  });
}

int lrp_train_get_master(lrp_handle* h, float* flat_dev, void* stream) {
  return with_handle(h, [&]() -> int {
    if (!h || !flat_dev) return fail(LRP_ERR_INVALID, "null argument");
    if (!h->trainer.ready) return fail(LRP_ERR_STATE, "lrp_train_begin must run first");
    LRP_HIP_CHECK(hipMemcpyAsync(flat_dev, h->trainer.master.p, h->trainer.n_total * sizeof(float), hipMemcpyDeviceToDevice, S(stream)));
    return LRP_OK;
  });
}

int lrp_heatmap_scores(const float* R_img_dev, double* scores_dev, int32_t n, int32_t npix, int32_t C, int32_t mode,
                       void* stream) {
  return guarded([&]() -> int {
    if (!R_img_dev || !scores_dev) return fail(LRP_ERR_INVALID, "null argument");
    if (n < 1 || npix < 1 || C < 1) return fail(LRP_ERR_INVALID, "n, npix, C must be positive");
    if (mode < 0 || mode > 2) return fail(LRP_ERR_UNSUPPORTED, "the lrp inference mode is not available");
    hipLaunchKernelGGL(heatmap_score_kernel, dim3(n), dim3(256), 0, S(stream), R_img_dev, scores_dev, npix, C, mode);
    LRP_HIP_CHECK(hipGetLastError());
    return LRP_OK;
  });
}

int lrp_op_log_softmax_topk(const double* logits_dev, int32_t rows, int32_t V, int32_t k, int32_t* ids_dev, double* logp_dev,
                            void* stream) {
  return guarded([&]() -> int {
    if (!logits_dev || !ids_dev || !logp_dev) return fail(LRP_ERR_INVALID, "null argument");
    if (rows < 1 || V < 1 || k < 1 || k > V || k > TOPK_MAX) return fail(LRP_ERR_INVALID, "need rows >= 1 and 1 <= k <= min(V, %d)", TOPK_MAX);
    hipLaunchKernelGGL(log_softmax_topk_kernel, dim3(rows), dim3(256), 0, S(stream), logits_dev, V, k, ids_dev, logp_dev);
    LRP_HIP_CHECK(hipGetLastError());
    return LRP_OK;
  });
}

}  // extern "C"
