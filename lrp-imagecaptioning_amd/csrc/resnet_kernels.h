// resnet_kernels.h — streaming (HBM-bound) helpers of the ResNet-101 encoder path (config 4):
// BN + gate construction, block outputs / Add-rule factors, stride-2 gather / scatter, the
// overlapping 3x3/2 max-pool and its relevance routing, the 7x7/2 stem im2col and stencil.
// Rules: BatchNormalizationReverseLayer RA:197-257, AddReverseLayer RA:260-286, Alpha1Beta0 RR:274-322,
// gradient routing RA:470-480; architecture: keras_applications.resnet_common (ResNet v1 bottleneck).
#pragma once
#include <hip/hip_runtime.h>
#include "cnn_kernels.h"

namespace lrp {

__device__ __forceinline__ float stab_sign(float d) { return d + (d >= 0.f ? 1e-7f : -1e-7f); }

// max|.| of what a thread wrote -> one atomic per wave, spread over ACT_MAX_SLOTS float-bit slots (the fp16-pair forward
// conv of the next unit scales its input by a power of two taken from them, cnn_kernels.h: split_h_scaled_kernel)
__device__ __forceinline__ void rn_flush_max(float m, unsigned* __restrict__ slots) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0 && m > 0.f)
    atomicMax(slots + ((blockIdx.x + (threadIdx.x >> 6)) & (ACT_MAX_SLOTS - 1)), __float_as_uint(m));
}

// After a conv unit: c = conv(x,w)+b (exact), Z = alpha1beta0 denominator.
//   y = BN(c);  Q = c (y - beta) / stab((c - mu) y) / safe(Z)     [BN reverse o conv denominator]
//   relu != 0: act = relu(y), gate = act * Q  (what the NEXT conv's relevance is multiplied with)
//   relu == 0: act = y (pre-Add tensor), gate = Q
__global__ __launch_bounds__(256) void rn_bn_unit_kernel(const float* __restrict__ c, const float* __restrict__ Z,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const float* __restrict__ mean, const float* __restrict__ var,
                                                         float bn_eps, float* __restrict__ act, float* __restrict__ gate,
                                                         float* __restrict__ qonly, size_t n, int C, int relu,
                                                         unsigned* __restrict__ max_slots) {
  float m = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int ch = (int)(i % C);
    const float cv = c[i], mu = mean[ch], bt = beta[ch];
    const float y = gamma[ch] * (cv - mu) / sqrtf(var[ch] + bn_eps) + bt;
    const float q = (cv * (y - bt)) / safe_den(stab_sign((cv - mu) * y)) / safe_den(Z[i]);
    const float av = relu ? fmaxf(y, 0.f) : y;
    act[i] = av;
    m = fmaxf(m, fabsf(av));
    gate[i] = relu ? av * q : q;
    if (qonly) qonly[i] = q;
  }
  if (max_slots) rn_flush_max(m, max_slots);
}

// Block end: o = relu(sc + y3);  GA = y3 / safe(sc + y3) * Q3;  GS = sc / safe(sc + y3) [* Q0 for a projection shortcut]
__global__ __launch_bounds__(256) void rn_block_out_kernel(const float* __restrict__ sc, const float* __restrict__ y3,
                                                           const float* __restrict__ Q3, const float* __restrict__ Q0,
                                                           float* __restrict__ o, float* __restrict__ GA,
                                                           float* __restrict__ GS, size_t n,
                                                           unsigned* __restrict__ max_slots) {
  // 16 B per lane and tensor (n is a multiple of 4: channel counts are)
  float m = 0.f;
  const size_t n4 = n / 4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const f32x4 s = reinterpret_cast<const f32x4*>(sc)[i], y = reinterpret_cast<const f32x4*>(y3)[i];
    const f32x4 q3 = reinterpret_cast<const f32x4*>(Q3)[i];
    f32x4 q0 = {1.f, 1.f, 1.f, 1.f};
    if (Q0) q0 = reinterpret_cast<const f32x4*>(Q0)[i];
    f32x4 ov, ga, gs;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float den = safe_den(s[q] + y[q]);
      ov[q] = fmaxf(s[q] + y[q], 0.f);
      m = fmaxf(m, s[q] + y[q]);
      ga[q] = y[q] / den * q3[q];
      gs[q] = Q0 ? s[q] / den * q0[q] : s[q] / den;
    }
    reinterpret_cast<f32x4*>(o)[i] = ov;
    reinterpret_cast<f32x4*>(GA)[i] = ga;
    reinterpret_cast<f32x4*>(GS)[i] = gs;
  }
  if (max_slots) rn_flush_max(m, max_slots);
}

// ---- round 4: the fp16-pair forward without passes between the convs (the VGG encoder's round-3 design) ----
// Every conv unit's epilogue writes its activation as the NEXT conv's operand (pairs scaled by a power of two); the block-end
// kernel and the stem's pool do the same for the block inputs.  A producer's scale has to be known before it runs, so it
// comes from a bound, and the bounds of a whole bottleneck block hang on ONE measured quantity — the maximum of the block's
// input, per image (an image's result does not depend on its batch mates):
//   block input t:   |sc + y3| <= max|sc| + max|y3|  (both measured: the producing convs raise max-slots; within 2x of the true
//                    maximum)                         stem pool: max of the pooled map = max of the map (measured)
//   conv + BN unit:  |y_c| <= |gamma_c| / sd_c * (max|x| * sum_k |w_c[k]| + |b_c - mean_c|) + |beta_c|   ->  norm = {A, B},
//                    bound(a1) = bound(t) * A1 + B1,  bound(a2) = bound(a1) * A2 + B2     (rn_unit_norm_kernel; relu(y) <= |y|)
// so the kernel that PRODUCES a block input derives every scale of the block that reads it (RnBlockScales below) and no scale
// launch runs between the convs.  [MI355X: as one tiny launch in front of every producer, 99 per encode, they cost 0.49 of the
// 1.30 ms the split passes had cost.]  The bounds overshoot — a row sum against a typical dot product: 20-100x per level, two
// levels deep, then the next block starts from a measured maximum again — and what an overshoot costs is the fp16 subnormal floor
// moving up: absolute error <= 2^-25 * overshoot / 30000 of the tensor's maximum (cnn_kernels.h, fwd_scale_kernel), 1e-8 at 10^4.
struct RnBlockScales {
  const float* norm1; const float* norm2;                // {A, B} of the reading block's units 1 and 2
  const float* wsc1; const float* wsc0; const float* wsc2; const float* wsc3;   // weight scale records of its units (wsc0: projection or null)
  float* osc_t; float* osc1; float* osc2;                // [images] pair scales: block input, a1, a2
  float* us1; float* us0; float* us2; float* us3;        // [images] input unscales of units 1, 0 (or null), 2, 3
};
__device__ __forceinline__ int rn_pow2_scale(float bound) {   // k with bound * 2^k in (15000, 30000]
  int k = 0;
  if (bound > 0.f && bound < 3.0e38f) {
    k = (int)floorf(log2f(30000.f / bound));
    k = k < -120 ? -120 : k > 120 ? 120 : k;
  }
  return k;
}
// every thread of the block calls it: returns the pair scale of the block input of image n (bound_t from the image's slots:
// max(slotsA) + max(slotsB)); `write`: this thread also stores the derived scales of the reading block
__device__ __forceinline__ float rn_block_scales(const unsigned* __restrict__ slotsA, const unsigned* __restrict__ slotsB, int n,
                                                 const RnBlockScales& S, bool write) {
  const int l = threadIdx.x & (ACT_MAX_SLOTS - 1);
  float m = __uint_as_float(slotsA[(size_t)n * ACT_MAX_SLOTS + l]);
  float mb = slotsB ? __uint_as_float(slotsB[(size_t)n * ACT_MAX_SLOTS + l]) : 0.f;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    m = fmaxf(m, __shfl_xor(m, o));
    mb = fmaxf(mb, __shfl_xor(mb, o));
  }
  const float bt = m + mb;
  const int kt = rn_pow2_scale(bt);
  if (write) {
    const float b1 = bt * S.norm1[0] + S.norm1[1];
    const float b2 = b1 * S.norm2[0] + S.norm2[1];
    const int k1 = rn_pow2_scale(b1), k2 = rn_pow2_scale(b2);
    S.osc_t[n] = ldexpf(1.f, kt); S.osc1[n] = ldexpf(1.f, k1); S.osc2[n] = ldexpf(1.f, k2);
    S.us1[n] = ldexpf(1.f, -kt) * S.wsc1[1];
    if (S.us0) S.us0[n] = ldexpf(1.f, -kt) * S.wsc0[1];
    S.us2[n] = ldexpf(1.f, -k1) * S.wsc2[1];
    S.us3[n] = ldexpf(1.f, -k2) * S.wsc3[1];
  }
  return ldexpf(1.f, kt);
}
// norm = {A, B} of a conv + BN unit (above); w = the unit's packed forward matrix [rows >= cout][K] (zero padded), one block
// per output channel; norm zeroed before (non-negative floats order like their bit patterns)
__global__ __launch_bounds__(256) void rn_unit_norm_kernel(const float* __restrict__ w, int K, const float* __restrict__ b,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ mean, const float* __restrict__ var,
                                                           float bn_eps, float* __restrict__ norm) {
  __shared__ float red[256];
  const int c = blockIdx.x;
  const float* r = w + (size_t)c * K;
  float s = 0.f;
  for (int k = threadIdx.x; k < K; k += 256) s += fabsf(r[k]);
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float g = fabsf(gamma[c]) / sqrtf(var[c] + bn_eps);
    atomicMax(reinterpret_cast<unsigned*>(norm), __float_as_uint(g * red[0] * 1.0001f));
    atomicMax(reinterpret_cast<unsigned*>(norm) + 1, __float_as_uint((g * fabsf(b[c] - mean[c]) + fabsf(beta[c])) * 1.0001f));
  }
}
// rn_bn_unit_kernel per image: grid (blocks per image, images), max|act| into slots[image][ACT_MAX_SLOTS]  (the stem)
__global__ __launch_bounds__(256) void rn_bn_unit_img_kernel(const float* __restrict__ c, const float* __restrict__ Z,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ mean, const float* __restrict__ var,
                                                             float bn_eps, float* __restrict__ act, float* __restrict__ gate,
                                                             float* __restrict__ qonly, size_t per_img, int C, int relu,
                                                             unsigned* __restrict__ slots) {
  const size_t base = (size_t)blockIdx.y * per_img;
  float m = 0.f;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < per_img; e += (size_t)gridDim.x * 256) {
    const size_t i = base + e;
    const int ch = (int)(e % C);
    const float cv = c[i], mu = mean[ch], bt = beta[ch];
    const float y = gamma[ch] * (cv - mu) / sqrtf(var[ch] + bn_eps) + bt;
    const float q = (cv * (y - bt)) / safe_den(stab_sign((cv - mu) * y)) / safe_den(Z[i]);
    const float av = relu ? fmaxf(y, 0.f) : y;
    act[i] = av;
    m = fmaxf(m, fabsf(av));
    gate[i] = relu ? av * q : q;
    if (qonly) qonly[i] = q;
  }
  rn_flush_max(m, slots + (size_t)blockIdx.y * ACT_MAX_SLOTS);
}
// rn_block_out_kernel per image, eight channels per thread, and the block output also as the next block's fp16 pairs, with the
// scales of that block derived here (rn_block_scales: slots_y3 / slots_sc = the max-slots of the two summands); grid (blocks
// per image, images); slots_out raised to max(o).  Last block of the network: pairs == nullptr, nothing but o / GA / GS.
__global__ __launch_bounds__(256) void rn_block_out_pairs_kernel(const float* __restrict__ sc, const float* __restrict__ y3,
                                                                 const float* __restrict__ Q3, const float* __restrict__ Q0,
                                                                 float* __restrict__ o, float* __restrict__ GA, float* __restrict__ GS,
                                                                 size_t per_img8, unsigned* __restrict__ slots_out,
                                                                 float* __restrict__ pairs, const unsigned* __restrict__ slots_y3,
                                                                 const unsigned* __restrict__ slots_sc, RnBlockScales S) {
  const int n = blockIdx.y;
  const size_t base = (size_t)n * per_img8;
  const float ps = pairs ? rn_block_scales(slots_y3, slots_sc, n, S, blockIdx.x == 0 && threadIdx.x == 0) : 1.f;
  float m = 0.f;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < per_img8; e += (size_t)gridDim.x * 256) {
    const size_t i = (base + e) * 8;
    float ov[8];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const f32x4 s = *reinterpret_cast<const f32x4*>(sc + i + 4 * g), y = *reinterpret_cast<const f32x4*>(y3 + i + 4 * g);
      const f32x4 q3 = *reinterpret_cast<const f32x4*>(Q3 + i + 4 * g);
      f32x4 q0 = {1.f, 1.f, 1.f, 1.f};
      if (Q0) q0 = *reinterpret_cast<const f32x4*>(Q0 + i + 4 * g);
      f32x4 o4, ga, gs;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float den = safe_den(s[q] + y[q]);
        o4[q] = fmaxf(s[q] + y[q], 0.f);
        m = fmaxf(m, s[q] + y[q]);
        ga[q] = y[q] / den * q3[q];
        gs[q] = Q0 ? s[q] / den * q0[q] : s[q] / den;
        ov[4 * g + q] = o4[q] * ps;
      }
      *reinterpret_cast<f32x4*>(o + i + 4 * g) = o4;
      *reinterpret_cast<f32x4*>(GA + i + 4 * g) = ga;
      *reinterpret_cast<f32x4*>(GS + i + 4 * g) = gs;
    }
    if (pairs) split8h_store(ov, pairs + i);
  }
  if (slots_out) rn_flush_max(m, slots_out + (size_t)n * ACT_MAX_SLOTS);
}

// the same product written in the split-bf16 operand format of the conv kernel (8 channels per thread; head of a
// conv chain in the bf16x3 mode)
__global__ __launch_bounds__(256) void rn_mul_gate_split_kernel(const float* __restrict__ R, const float* __restrict__ G,
                                                                const int* __restrict__ row2img, float* __restrict__ out,
                                                                int ntok, size_t per_img8) {
  const size_t total = (size_t)ntok * per_img8;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int t = (int)(i / per_img8);
    const size_t e = i - (size_t)t * per_img8;
    const int img = row2img ? row2img[t] : t;
    const float* r = R + i * 8;
    const float* g = G + ((size_t)img * per_img8 + e) * 8;
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = r[q] * g[q];
    split8_store(v, out + i * 8);
  }
}

// out[t][e] = R[t][e] * G[img(t)][e]  (+ add[t][e])
__global__ __launch_bounds__(256) void rn_mul_gate_kernel(const float* __restrict__ R, const float* __restrict__ G,
                                                          const int* __restrict__ row2img, const float* __restrict__ add,
                                                          float* __restrict__ out, int ntok, size_t per_img) {
  const size_t total = (size_t)ntok * per_img;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int t = (int)(i / per_img);
    const size_t e = i - (size_t)t * per_img;
    const int img = row2img ? row2img[t] : t;
    float v = R[i] * G[(size_t)img * per_img + e];
    if (add) v += add[i];
    out[i] = v;
  }
}

__global__ __launch_bounds__(256) void rn_add_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                     float* __restrict__ out, size_t n) {
  const size_t n4 = n / 4;                               // (element counts are multiples of the channel count, itself % 4 == 0)
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256)
    reinterpret_cast<f32x4*>(out)[i] = reinterpret_cast<const f32x4*>(a)[i] + reinterpret_cast<const f32x4*>(b)[i];
}

// stride-2 gather: xs[n][ho][wo][c] = x[n][2ho][2wo][c]   (input of a strided 1x1 conv); four channels per thread
__global__ __launch_bounds__(256) void rn_subsample2_kernel(const float* __restrict__ x, float* __restrict__ xs, int NB,
                                                            int H, int W, int C) {
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2, C4 = C / 4;
  const size_t total = (size_t)NB * Ho * Wo * C4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % C4) * 4;
    size_t r = i / C4;
    const int wo = (int)(r % Wo);
    r /= Wo;
    const int ho = (int)(r % Ho);
    const int n = (int)(r / Ho);
    reinterpret_cast<f32x4*>(xs)[i] = *reinterpret_cast<const f32x4*>(x + (((size_t)n * H + 2 * ho) * W + 2 * wo) * C + c);
  }
}

// stride-2 scatter of relevance: fine[n][h][w][c] = (h,w both even) ? (a[..] (+ b[..])) : 0; four channels per thread
__global__ __launch_bounds__(256) void rn_scatter2_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          float* __restrict__ fine, int NB, int H, int W, int C) {
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2, C4 = C / 4;
  const size_t total = (size_t)NB * H * W * C4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % C4) * 4;
    size_t r = i / C4;
    const int w = (int)(r % W);
    r /= W;
    const int h = (int)(r % H);
    const int n = (int)(r / H);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (!(h & 1) && !(w & 1)) {
      const size_t j = (((size_t)n * Ho + (h >> 1)) * Wo + (w >> 1)) * C + c;
      v = *reinterpret_cast<const f32x4*>(a + j);
      if (b) v += *reinterpret_cast<const f32x4*>(b + j);
    }
    reinterpret_cast<f32x4*>(fine)[i] = v;
  }
}

// End of a projection block's walk in ONE pass (bf16x3 chains, C % 8 == 0; eight channels per thread):
//   fine = a + b   (SCATTER: at the even positions of the stride-2 grid, zeros elsewhere — the two kernels above in one)
//   out2s = split8(fine * G2[img])   (optional: S3 = R_t * GA of the block walked NEXT, which would otherwise re-read R_t)
template <bool SCATTER>
__global__ __launch_bounds__(256) void rn_join_split_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                            float* __restrict__ fine, const float* __restrict__ G2,
                                                            const int* __restrict__ row2img, float* __restrict__ out2s, int NB,
                                                            int H, int W, int C) {
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2, C8 = C / 8;
  const size_t total = (size_t)NB * H * W * C8;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % C8) * 8;
    size_t r = i / C8;
    const int w = (int)(r % W);
    r /= W;
    const int h = (int)(r % H);
    const int n = (int)(r / H);
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = 0.f;
    if (!SCATTER || (!(h & 1) && !(w & 1))) {
      const size_t j = SCATTER ? (((size_t)n * Ho + (h >> 1)) * Wo + (w >> 1)) * C + c : i * 8;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const f32x4 s = *reinterpret_cast<const f32x4*>(a + j + 4 * g) + *reinterpret_cast<const f32x4*>(b + j + 4 * g);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[4 * g + q] = s[q];
      }
    }
    *reinterpret_cast<f32x4*>(fine + i * 8) = *reinterpret_cast<const f32x4*>(v);
    *reinterpret_cast<f32x4*>(fine + i * 8 + 4) = *reinterpret_cast<const f32x4*>(v + 4);
    if (out2s) {
      const int img = row2img ? row2img[n] : n;
      const float* g2 = G2 + (((size_t)img * H + h) * W + w) * C + c;
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(g2), g1 = *reinterpret_cast<const f32x4*>(g2 + 4);
#pragma unroll
      for (int q = 0; q < 4; ++q) { v[q] *= g0[q]; v[4 + q] *= g1[q]; }
      split8_store(v, out2s + i * 8);
    }
  }
}

// 3x3/2 max-pool on the 1-padded map (ZeroPadding2D(1) + MaxPooling2D(3, 2, 'valid')):  a (H,W) -> (H/2, W/2)
__device__ __forceinline__ float rn_padded_at(const float* __restrict__ a, int n, int i, int j, int c, int H, int W, int C) {
  return (i >= 0 && i < H && j >= 0 && j < W) ? a[(((size_t)n * H + i) * W + j) * C + c] : 0.f;   // zero padding is a candidate
}
// win[i] = kh*3 + kw of the FIRST maximum in window scan order (what tf.gradients routes to): the per-token routing
// reads this byte instead of re-deriving the arg-max of up to four windows (36 loads) for every token
__global__ __launch_bounds__(256) void rn_pool3_kernel(const float* __restrict__ a, float* __restrict__ out,
                                                       unsigned char* __restrict__ win, int NB, int H, int W, int C) {
  const int Ho = H / 2, Wo = W / 2;
  const size_t total = (size_t)NB * Ho * Wo * C;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    size_t r = i / C;
    const int ow = (int)(r % Wo);
    r /= Wo;
    const int oh = (int)(r % Ho);
    const int n = (int)(r / Ho);
    float m = -INFINITY;
    int arg = 0;
    for (int kh = 0; kh < 3; ++kh)
      for (int kw = 0; kw < 3; ++kw) {
        const float v = rn_padded_at(a, n, 2 * oh + kh - 1, 2 * ow + kw - 1, c, H, W, C);
        if (v > m) { m = v; arg = kh * 3 + kw; }
      }
    out[i] = m;
    win[i] = (unsigned char)arg;
  }
}

// the same pool, eight channels per thread, its output also as the first block's fp16 pairs, that block's scales derived here
// from the stem activation's max-slots (rn_block_scales); grid (blocks per image, images).  (a >= 0 behind the stem's ReLU, so the zero padding never wins over a positive value and
// ties with it resolve in scan order like above.)
__global__ __launch_bounds__(256) void rn_pool3_pairs_kernel(const float* __restrict__ a, float* __restrict__ out,
                                                             unsigned char* __restrict__ win, float* __restrict__ pairs,
                                                             const unsigned* __restrict__ slots_a, RnBlockScales S, int H, int W, int C) {
  const int Ho = H / 2, Wo = W / 2, C8 = C / 8, n = blockIdx.y;
  const size_t per_img8 = (size_t)Ho * Wo * C8;
  const float ps = rn_block_scales(slots_a, nullptr, n, S, blockIdx.x == 0 && threadIdx.x == 0);
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < per_img8; e += (size_t)gridDim.x * 256) {
    const int c = (int)(e % C8) * 8;
    size_t r = e / C8;
    const int ow = (int)(r % Wo);
    const int oh = (int)(r / Wo);
    float m[8];
    unsigned arg[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { m[q] = -INFINITY; arg[q] = 0; }
    for (int kh = 0; kh < 3; ++kh)
      for (int kw = 0; kw < 3; ++kw) {
        const int i = 2 * oh + kh - 1, j = 2 * ow + kw - 1;
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = 0.f;                                 // zero padding is a candidate
        if (i >= 0 && i < H && j >= 0 && j < W) {
          const float* p = a + (((size_t)n * H + i) * W + j) * C + c;
          *reinterpret_cast<f32x4*>(v) = *reinterpret_cast<const f32x4*>(p);
          *reinterpret_cast<f32x4*>(v + 4) = *reinterpret_cast<const f32x4*>(p + 4);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (v[q] > m[q]) { m[q] = v[q]; arg[q] = (unsigned)(kh * 3 + kw); }
      }
    const size_t o = ((size_t)n * per_img8 + e) * 8;
    *reinterpret_cast<f32x4*>(out + o) = *reinterpret_cast<const f32x4*>(m);
    *reinterpret_cast<f32x4*>(out + o + 4) = *reinterpret_cast<const f32x4*>(m + 4);
    typedef unsigned u32x2p __attribute__((ext_vector_type(2)));
    u32x2p w8;
    w8[0] = arg[0] | (arg[1] << 8) | (arg[2] << 16) | (arg[3] << 24);
    w8[1] = arg[4] | (arg[5] << 8) | (arg[6] << 16) | (arg[7] << 24);
    *reinterpret_cast<u32x2p*>(win + o) = w8;
#pragma unroll
    for (int q = 0; q < 8; ++q) m[q] *= ps;
    split8h_store(m, pairs + o);
  }
}

// Relevance routing through that pool (gradient of max-pool = first arg-max of each window, windows overlap),
// fused with the stem gate:  S[n][i][j][c] = Q[img][i][j][c] * sum_{windows (oh,ow) whose arg-max is (i,j)} R[n][oh][ow][c]
// SPLIT: eight channels per thread, written in the split-bf16 operand format of the conv kernel (the stem's tap GEMM then
// runs as bf16x3); otherwise four channels per thread, plain fp32.
template <bool SPLIT>
__global__ __launch_bounds__(256) void rn_pool3_route_kernel(const float* __restrict__ R, const unsigned char* __restrict__ win,
                                                             const float* __restrict__ Q, const int* __restrict__ row2img,
                                                             float* __restrict__ S, int ntok, int H, int W, int C) {
  constexpr int CW = SPLIT ? 8 : 4;                      // (C % CW == 0): 16 B relevance / gate loads, 4 B of winners per window
  const int Ho = H / 2, Wo = W / 2, CG = C / CW;
  const size_t total = (size_t)ntok * H * W * CG;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int c = (int)(idx % CG) * CW;
    size_t r = idx / CG;
    const int j = (int)(r % W);
    r /= W;
    const int i = (int)(r % H);
    const int t = (int)(r / H);
    const int img = row2img ? row2img[t] : t;
    float acc[CW];
#pragma unroll
    for (int q = 0; q < CW; ++q) acc[q] = 0.f;
    // windows containing padded position (i+1, j+1): oh in [ceil((i-1)/2), floor((i+1)/2)]
    for (int oh = (i) / 2; oh <= (i + 1) / 2; ++oh) {
      if (oh < 0 || oh >= Ho || 2 * oh - 1 > i || 2 * oh + 1 < i) continue;
      for (int ow = (j) / 2; ow <= (j + 1) / 2; ++ow) {
        if (ow < 0 || ow >= Wo || 2 * ow - 1 > j || 2 * ow + 1 < j) continue;
        // (i, j) is position (kh, kw) = (i - 2 oh + 1, j - 2 ow + 1) of this window: did it win?
        const unsigned mine = (unsigned)((i - 2 * oh + 1) * 3 + (j - 2 * ow + 1));
        const size_t wo = (((size_t)img * Ho + oh) * Wo + ow) * C + c, ro = (((size_t)t * Ho + oh) * Wo + ow) * C + c;
#pragma unroll
        for (int g = 0; g < CW / 4; ++g) {
          const unsigned w4 = *reinterpret_cast<const unsigned*>(win + wo + 4 * g);
          const f32x4 rv = *reinterpret_cast<const f32x4*>(R + ro + 4 * g);
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (((w4 >> (8 * q)) & 0xFFu) == mine) acc[4 * g + q] += rv[q];
        }
      }
    }
    const float* qp = Q + (((size_t)img * H + i) * W + j) * C + c;
#pragma unroll
    for (int g = 0; g < CW / 4; ++g) {
      const f32x4 qv = *reinterpret_cast<const f32x4*>(qp + 4 * g);
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[4 * g + q] *= qv[q];
    }
    if constexpr (SPLIT) split8_store(acc, S + idx * 8);
    else *reinterpret_cast<f32x4*>(S + idx * 4) = *reinterpret_cast<const f32x4*>(acc);
  }
}

// Stem forward as a 1-tap GEMM: A[m][320] = [ x+ patch (7*7*3 = 147) | 0*13 | x- patch (147) | 0*13 ] for output pixel
// m = (n, oh, ow); patch element (kh,kw,c) = x[2oh+kh-3][2ow+kw-3][c] (ZeroPadding2D(3) + 7x7/2 'valid')
constexpr int RN_STEM_K = 160;
__global__ __launch_bounds__(256) void rn_stem_im2col_kernel(const float* __restrict__ img, float* __restrict__ A, int NB,
                                                             int H, int W) {
  const int Ho = H / 2, Wo = W / 2;
  const size_t total = (size_t)NB * Ho * Wo * (2 * RN_STEM_K);
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int k = (int)(idx % (2 * RN_STEM_K));
    size_t m = idx / (2 * RN_STEM_K);
    const int ow = (int)(m % Wo);
    m /= Wo;
    const int oh = (int)(m % Ho);
    const int n = (int)(m / Ho);
    const int kk = k % RN_STEM_K;
    float v = 0.f;
    if (kk < 147) {
      const int tap = kk / 3, c = kk - 3 * tap;
      const int i = 2 * oh + tap / 7 - 3, j = 2 * ow + tap % 7 - 3;
      if (i >= 0 && i < H && j >= 0 && j < W) {
        const float x = img[(((size_t)n * H + i) * W + j) * 3 + c];
        v = (k < RN_STEM_K) ? (x >= 0.f ? x : 0.f) : (x < 0.f ? x : 0.f);
      }
    }
    A[idx] = v;
  }
}

// Stem reverse: T[q][tap*6 + c] = sum_co S[q][co] w+[tap][c][co] (c<3) / w-[..] (c>=3) came from one K = C_stem GEMM;
// R_img[p][c] = x+[p][c] * sum_{tap} T+[q(p,tap)][tap][c] + x-[p][c] * sum T-[...],  q = ((i+3-kh)/2, (j+3-kw)/2) when integral
constexpr int RN_STEM_TCOLS = 294;

// Device twin of RnEncoder::pack_unit's stem branch (lrp_set_weight_dev: the weights arrive in HBM and stay there).
// w: HWIO (7,7,3,cout).  wa / wz [Np][2 * RN_STEM_K]: a rows hold w against the x+ and the x- patch, z rows w+ / w-;
// wb [Npb][Kb]: row t*6 + c = w+[t][c][:], row t*6 + 3 + c = w-[t][c][:] (the tap GEMM of the walk).  All zero padded.
__global__ __launch_bounds__(256) void rn_pack_stem_dev_kernel(const float* __restrict__ w, float* __restrict__ wa, float* __restrict__ wz,
                                                               float* __restrict__ wb, int cout, int Np, int Npb, int Kb) {
  const int K = 2 * RN_STEM_K;
  const size_t nf = (size_t)Np * K, nb = (size_t)Npb * Kb;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nf + nb; i += (size_t)gridDim.x * 256) {
    if (i < nf) {
      const int co = (int)(i / K), k = (int)(i % K), kk = k % RN_STEM_K, neg = k / RN_STEM_K;
      float va = 0.f, vz = 0.f;
      if (co < cout && kk < 147) {
        const float v = w[(size_t)kk * cout + co];
        va = v;
        vz = neg ? (v < 0.f ? v : 0.f) : (v >= 0.f ? v : 0.f);
      }
      wa[i] = va; wz[i] = vz;
    } else {
      const size_t q = i - nf;
      const int row = (int)(q / Kb), co = (int)(q % Kb), t = row / 6, c6 = row % 6, c = c6 % 3;
      float v = 0.f;
      if (t < 49 && co < cout) {
        const float x = w[((size_t)t * 3 + c) * cout + co];
        v = c6 < 3 ? (x >= 0.f ? x : 0.f) : (x < 0.f ? x : 0.f);
      }
      wb[q] = v;
    }
  }
}
// ---- the reverse of the 7x7 / stride-2 stem in ONE launch (round 4) -------------------------------------------------------------
// Unfused (below: a 1-tap GEMM T = S . W with 49 taps x 6 columns per stem position, then rn_stem_stencil_kernel gathering from
// T) the T tensor — 4.7 GB at 320 tokens — is written and read back: 3.0 of config 4's 30 ms walk.  Here a workgroup owns a
// 16 x 16 PATCH of stem positions: their S rows (64 channels, split-bf16 pairs or fp32) stay in LDS, the 294 columns are
// produced ten taps (60 columns) at a time on the matrix cores into an LDS tile, and every thread adds the taps that land on ITS
// input pixels into registers — pixel (y, x) receives tap (kh, kw) from stem position ((y + 3 - kh) / 2, (x + 3 - kw) / 2) where
// both are whole — before the next ten taps overwrite the tile.  Patches step 13 positions: the 26 x 26 input pixels whose whole
// 4 x 4 neighbourhood of stem positions lies inside a patch are that patch's (1.5 x recompute of a GEMM that is 2 % of the walk).
// S is read 1.5 x, R_img written once; nothing else moves.  8 waves; wave w owns patch rows 32 w .. 32 w + 31 of the GEMM and keeps
// them in registers.  [MI355X, 320 tokens: 3.05 ms (GEMM 1.56 + gather 1.49) -> see DESIGN 4.8; with the S rows in LDS — one
// workgroup per CU, every phase behind a barrier — the launch took 2.2 ms.]
constexpr int RN_SP = 16, RN_ST = 13, RN_SO = 2 * RN_ST;      // patch edge, patch step (stem positions), output pixels per patch edge
constexpr int RN_SROW = 68;                                   // LDS row pitch of the W tile in floats (64 + 4: conflict-free 16 B reads)
constexpr int RN_STS = 61;                                    // ... of the T tile (60 columns used; odd: the stencil's column reads spread over the banks)
constexpr int RN_STEM_LDS = (64 * RN_SROW + 256 * RN_STS) * 4;   // 79.9 KB: two workgroups per CU
template <bool SPLIT>
__global__ __launch_bounds__(512, 2) void rn_stem_reverse_kernel(const float* __restrict__ S, const float* __restrict__ Wb,
                                                                 const float* __restrict__ ximg, const int* __restrict__ row2img,
                                                                 float* __restrict__ out, int Ho, int Wo, int tiles_x, int tiles_y) {
  extern __shared__ __attribute__((aligned(16))) float rn_smem[];
  float* Bs = rn_smem;                                        // [64 columns of this tap group][RN_SROW]
  float* Ts = Bs + 64 * RN_SROW;                              // [256][RN_STS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tpi = tiles_x * tiles_y;
  const int pn = blockIdx.x / tpi, rr = blockIdx.x - pn * tpi, ty = rr / tiles_x, tx = rr - ty * tiles_x;
  const int oh0 = ty * RN_ST - 1, ow0 = tx * RN_ST - 1;       // first stem position of the patch
  const int H2 = 2 * Ho, W2 = 2 * Wo;
  const int h = lane >> 5, l31 = lane & 31;
  // The wave's 32 patch rows as MFMA A fragments, straight from memory into registers (they are reused by all five tap groups;
  // in LDS they cost 68 KB and a second workgroup per CU): lane (row, half h) holds, per k-step, the 32 B of channel group
  // 2 ks + h (split8: hi8 | lo8) — or for fp32 the eight 16 B pieces 8 g + 4 h.  Positions outside the map are zero rows.
  u32x4 af[8];
  {
    const int row = wave * 32 + l31, oh = oh0 + (row >> 4), ow = ow0 + (row & 15);
    const bool ok = oh >= 0 && oh < Ho && ow >= 0 && ow < Wo;
    const float* sp = S + (((size_t)pn * Ho + (ok ? oh : 0)) * Wo + (ok ? ow : 0)) * 64;
    const u32x4 z4 = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float* g = SPLIT ? sp + (2 * q + h) * 8 : sp + 16 * q + 4 * h;      // fp32: g = 2 q, 2 q + 1
      const u32x4 v0 = *reinterpret_cast<const u32x4*>(g), v1 = *reinterpret_cast<const u32x4*>(g + (SPLIT ? 4 : 8));
      af[2 * q] = ok ? v0 : z4;
      af[2 * q + 1] = ok ? v1 : z4;
    }
  }
  // this thread's input pixels and their running sums
  float pos[2][3], neg[2][3];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int c = 0; c < 3; ++c) { pos[q][c] = 0.f; neg[q][c] = 0.f; }
  f32x4 bq[2];                                                // the next tap group's W rows (64 rows x 16 pieces = 2 per thread)
  auto load_b = [&](int grp) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int p = tid + 512 * i, row = p >> 4, c16 = p & 15;
      bq[i] = *reinterpret_cast<const f32x4*>(Wb + (size_t)(grp * 60 + row) * 64 + c16 * 4);
    }
  };
  load_b(0);
  for (int grp = 0; grp < 5; ++grp) {
    __syncthreads();                                          // the previous group's Bs / Ts have been read
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int p = tid + 512 * i;
      *reinterpret_cast<f32x4*>(Bs + (p >> 4) * RN_SROW + (p & 15) * 4) = bq[i];
    }
    __syncthreads();
    if (grp + 1 < 5) load_b(grp + 1);
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    if constexpr (SPLIT) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int go = (2 * ks + h) * 8;
        const bf16x8 ah = __builtin_bit_cast(bf16x8, af[2 * ks]), al = __builtin_bit_cast(bf16x8, af[2 * ks + 1]);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float* brow = Bs + (j * 32 + l31) * RN_SROW + go;
          const bf16x8 bh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(brow));
          const bf16x8 bl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(brow + 4));
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[j], 0, 0, 0);      // small terms first
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[j], 0, 0, 0);
        }
      }
    } else {
      // exact fp32: lane half h takes k = 8 g + 4 h + s at sub-step s (one 16 B read of W and four MFMAs)
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const f32x4 a4 = __builtin_bit_cast(f32x4, af[g]);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(Bs + (j * 32 + l31) * RN_SROW + 8 * g + 4 * h);
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[s4], b4[s4], acc[j], 0, 0, 0);
        }
      }
    }
    // C map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      if (j * 32 + l31 < 60) {
#pragma unroll
        for (int r = 0; r < 16; ++r) Ts[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * RN_STS + j * 32 + l31] = acc[j][r];
      }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int o = tid + 512 * q;
      if (o < RN_SO * RN_SO) {
        const int oy = o / RN_SO, ox = o - oy * RN_SO;
        const int y = 2 * (oh0 + 1) + oy, x = 2 * (ow0 + 1) + ox;
#pragma unroll
        for (int tl = 0; tl < 10; ++tl) {
          const int tap = grp * 10 + tl, kh = tap / 7, kw = tap - kh * 7;
          if (tap < 49 && !((y + 3 - kh) & 1) && !((x + 3 - kw) & 1)) {
            const float* r = Ts + ((((y + 3 - kh) >> 1) - oh0) * RN_SP + (((x + 3 - kw) >> 1) - ow0)) * RN_STS + tl * 6;
            pos[q][0] += r[0]; pos[q][1] += r[1]; pos[q][2] += r[2];
            neg[q][0] += r[3]; neg[q][1] += r[4]; neg[q][2] += r[5];
          }
        }
      }
    }
  }
  const int img = row2img ? row2img[pn] : pn;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int o = tid + 512 * q;
    if (o >= RN_SO * RN_SO) continue;
    const int oy = o / RN_SO, ox = o - oy * RN_SO;
    const int y = 2 * (oh0 + 1) + oy, x = 2 * (ow0 + 1) + ox;
    if (y >= H2 || x >= W2) continue;
    const size_t pix = (size_t)y * W2 + x;
    const float* xv = ximg + ((size_t)img * H2 * W2 + pix) * 3;
    float* ov = out + ((size_t)pn * H2 * W2 + pix) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) ov[c] = xv[c] >= 0.f ? xv[c] * pos[q][c] : xv[c] * neg[q][c];
  }
}

__global__ __launch_bounds__(256) void rn_stem_stencil_kernel(const float* __restrict__ T, const float* __restrict__ ximg,
                                                              const int* __restrict__ row2img, float* __restrict__ out,
                                                              int ntok, int H, int W) {
  const int Ho = H / 2, Wo = W / 2;
  const size_t total = (size_t)ntok * H * W;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int HW = H * W;
    const int t = (int)(idx / HW), pix = (int)(idx - (size_t)t * HW);
    const int i = pix / W, j = pix - i * W;
    float pos[3] = {0.f, 0.f, 0.f}, neg[3] = {0.f, 0.f, 0.f};
    for (int kh = (i + 3) & 1; kh < 7; kh += 2) {
      const int oh = (i + 3 - kh) >> 1;
      if (oh < 0 || oh >= Ho) continue;
      for (int kw = (j + 3) & 1; kw < 7; kw += 2) {
        const int ow = (j + 3 - kw) >> 1;
        if (ow < 0 || ow >= Wo) continue;
        const float* r = T + (((size_t)t * Ho + oh) * Wo + ow) * RN_STEM_TCOLS + (kh * 7 + kw) * 6;
#pragma unroll
        for (int c = 0; c < 3; ++c) { pos[c] += r[c]; neg[c] += r[3 + c]; }
      }
    }
    const int img = row2img ? row2img[t] : t;
    const float* x = ximg + ((size_t)img * HW + pix) * 3;
    float* o = out + idx * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = x[c] >= 0.f ? x[c] * pos[c] : x[c] * neg[c];
  }
}

}  // namespace lrp
