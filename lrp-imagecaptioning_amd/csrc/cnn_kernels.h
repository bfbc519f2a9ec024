// cnn_kernels.h — the HBM-bound helpers of the encoder half (everything that is
// not a convolution).  All are pure streaming kernels: 16 B per lane, coalesced.
#pragma once
#include <hip/hip_runtime.h>
#include "conv_igemm.h"

namespace lrp {

// SafeDivide denominator, innvestigate/layers.py:456-458:  b + [b == 0] * K.epsilon()
__device__ __forceinline__ float safe_den(float z) { return z + (z == 0.f ? 1e-7f : 0.f); }

// Image layer as a 1-tap GEMM: A1[m][64] = [ x+ patch(27) | 0*5 | x- patch(27) | 0*5 ]
// so that  a_1 = A1 . [w ; w]   and   Z_1 = A1 . [w+ ; w-]  (RR:279-288, both sign branches).
__global__ __launch_bounds__(256) void im2col_image_kernel(const float* __restrict__ img, float* __restrict__ A1,
                                                           int NB, int H, int W) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;          // (pixel, group of 8 k): two 16-byte stores per thread
  const size_t total = (size_t)NB * H * W * 8;
  if (idx >= total) return;
  const int g = (int)(idx & 7);
  const size_t m = idx >> 3;
  const int HW = H * W;
  const int n = (int)(m / HW), rem = (int)(m - (size_t)n * HW);
  const int h = rem / W, w = rem - h * W;
  const bool neg = g >= 4;                                 // groups 0-3: x+ patch (k 0..31), 4-7: x- patch (k 32..63)
  float v[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int kk = 8 * (g & 3) + q;
    float x = 0.f;
    if (kk < 27) {
      const int t = kk / 3, c = kk - 3 * t;
      const int hh = h + t / 3 - 1, ww = w + t % 3 - 1;
      if (hh >= 0 && hh < H && ww >= 0 && ww < W) x = img[(((size_t)n * H + hh) * W + ww) * 3 + c];
    }
    v[q] = neg ? (x < 0.f ? x : 0.f) : (x >= 0.f ? x : 0.f);
  }
  *reinterpret_cast<f32x4*>(A1 + idx * 8) = *reinterpret_cast<const f32x4*>(v);
  *reinterpret_cast<f32x4*>(A1 + idx * 8 + 4) = *reinterpret_cast<const f32x4*>(v + 4);
}

// No pool after layer l:  G_l = a_l / safe(Z_l)     (x_{l+1} = a_l)
// (a and g may be the same buffer: the overlapped encode keeps a_l in the gate's storage until the gate is due)
__global__ __launch_bounds__(256) void gate_kernel(const f32x4* a, const f32x4* __restrict__ z, f32x4* g, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const f32x4 av = a[i], zv = z[i];
    f32x4 o;
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = av[c] / safe_den(zv[c]);
    g[i] = o;
  }
}

// 2x2/2 max-pool after layer l:  x_{l+1} = pool(a_l);  G_l = [first arg-max of the window] * a_l / safe(Z_l)
// (MaxPooling2D relevance = gradient routing, RA:470-480 -> layers.py:138-157)
// xnext == nullptr: gate only (the pooled activations were produced earlier by maxpool2_kernel); a may alias g
// (a thread reads its whole 2x2 window before it writes any of it).
__global__ __launch_bounds__(256) void pool_gate_kernel(const float* a, const float* __restrict__ z,
                                                        float* __restrict__ xnext, float* g,
                                                        int NB, int H, int W, int C) {
  const int C4 = C >> 2, Ho = H >> 1, Wo = W >> 1;
  const size_t total = (size_t)NB * Ho * Wo * C4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    size_t r = i / C4;
    const int wo = (int)(r % Wo);
    r /= Wo;
    const int ho = (int)(r % Ho);
    const int n = (int)(r / Ho);
    f32x4 av[4], zv[4];
    size_t off[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      off[p] = ((((size_t)n * H + 2 * ho + (p >> 1)) * W + 2 * wo + (p & 1)) * C) + 4 * c4;
      av[p] = *reinterpret_cast<const f32x4*>(a + off[p]);
      zv[p] = *reinterpret_cast<const f32x4*>(z + off[p]);
    }
    f32x4 mx = av[0];
    int arg[4] = {0, 0, 0, 0};
#pragma unroll
    for (int p = 1; p < 4; ++p)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (av[p][c] > mx[c]) { mx[c] = av[p][c]; arg[c] = p; }
    if (xnext) *reinterpret_cast<f32x4*>(xnext + (((size_t)n * Ho + ho) * Wo + wo) * C + 4 * c4) = mx;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      f32x4 o;
#pragma unroll
      for (int c = 0; c < 4; ++c) o[c] = (arg[c] == p) ? av[p][c] / safe_den(zv[p][c]) : 0.f;
      *reinterpret_cast<f32x4*>(g + off[p]) = o;
    }
  }
}

// plain 2x2/2 max-pool (the forward chain of the overlapped encode; the arg-max mask is recomputed by pool_gate_kernel)
__global__ __launch_bounds__(256) void maxpool2_kernel(const float* __restrict__ a, float* __restrict__ xnext,
                                                       int NB, int H, int W, int C) {
  const int C4 = C >> 2, Ho = H >> 1, Wo = W >> 1;
  const size_t total = (size_t)NB * Ho * Wo * C4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    size_t r = i / C4;
    const int wo = (int)(r % Wo);
    r /= Wo;
    const int ho = (int)(r % Ho);
    const int n = (int)(r / Ho);
    const float* p0 = a + ((((size_t)n * H + 2 * ho) * W + 2 * wo) * C) + 4 * c4;
    f32x4 mx = *reinterpret_cast<const f32x4*>(p0);
    const f32x4 v1 = *reinterpret_cast<const f32x4*>(p0 + C), v2 = *reinterpret_cast<const f32x4*>(p0 + (size_t)W * C),
                v3 = *reinterpret_cast<const f32x4*>(p0 + (size_t)W * C + C);
#pragma unroll
    for (int c = 0; c < 4; ++c) mx[c] = fmaxf(fmaxf(mx[c], v1[c]), fmaxf(v2[c], v3[c]));
    *reinterpret_cast<f32x4*>(xnext + (((size_t)n * Ho + ho) * Wo + wo) * C + 4 * c4) = mx;
  }
}

// fp32 NHWC -> split8 copy (operand format of the bf16x3 convs), 8 channels per thread
__global__ __launch_bounds__(256) void split_copy_kernel(const float* __restrict__ x, float* __restrict__ xs, size_t n8) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    float v[8];
    *reinterpret_cast<f32x4*>(v) = *reinterpret_cast<const f32x4*>(x + i * 8);
    *reinterpret_cast<f32x4*>(v + 4) = *reinterpret_cast<const f32x4*>(x + i * 8 + 4);
    split8_store(v, xs + i * 8);
  }
}

// Device twins of the host weight packers (conv_igemm.h pack_conv_fwd / pack_conv_bwd / pack_frag64) for the fine-tune
// step, whose weights change on the device every iteration.  dst [rows_total][taps * CP], zero padded.
//   bwd = 0: row = output channel, k = tap * CP + ci       (dual: rows [Cout, 2 Cout) hold w+ of row - Cout)
//   bwd = 1: row = input channel,  k = tap' * CP + co with the taps flipped (transposed conv as a conv)
__global__ __launch_bounds__(256) void pack_conv_dev_kernel(const float* __restrict__ w, float* __restrict__ dst, int bwd, int Cin,
                                                            int Cout, int CP, int rows_total, int dual, int pos_only, int taps = 9) {
  const int K = taps * CP;
  const size_t total = (size_t)rows_total * K;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    int row = (int)(i / K);
    const int k = (int)(i % K), t = k / CP, c = k % CP;
    bool pos = pos_only != 0;
    float v = 0.f;
    if (!bwd) {
      if (dual == 2) {                                  // interleaved dual: blocks of 32 rows, [w | w+] of the same 32 channels
        const int blk = row >> 5;
        pos = (blk & 1) != 0;
        row = (blk >> 1) * 32 + (row & 31);
      } else
      if (dual && row >= Cout) { row -= Cout; pos = true; }
      if (row < Cout && c < Cin) v = w[((size_t)t * Cin + c) * Cout + row];
    } else {
      if (row < Cin && c < Cout) v = w[((size_t)(taps - 1 - t) * Cin + row) * Cout + c];
    }
    dst[i] = (pos && !(v >= 0.f)) ? 0.f : v;
  }
}
// ---- fp16-pair operands (conv_igemm.h PREC_F16X2).  A weight matrix is stored scaled by a power of two, 2^k with
// k = floor(log2(16384 / max|w|)): unscaled, the lo halves of VGG-sized weights (|w| ~ 1e-3, lo ~ 5e-7) fall into fp16's
// subnormals and the pair keeps 15 bits instead of 21 [MI355X: feature error of the forward 2.4e-6 -> see DESIGN 4.1].
// Per matrix a device record wsc[4] = { 2^k, 2^-k, max row sum of |hi| (scaled), k } that the consumers read.
__global__ void wscale_kernel(const unsigned* __restrict__ slots, float* __restrict__ wsc) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float m = 0.f;
  for (int i = 0; i < ACT_MAX_SLOTS; ++i) m = fmaxf(m, __uint_as_float(slots[i]));
  int k = 0;
  if (m > 0.f && m < 3.0e38f) {
    k = (int)floorf(log2f(16384.f / m));
    k = k < -40 ? -40 : k > 40 ? 40 : k;
  }
  wsc[0] = ldexpf(1.f, k); wsc[1] = ldexpf(1.f, -k); wsc[2] = 0.f; wsc[3] = (float)k;
}
// fp32 packed matrix -> fp16 pairs of (w * wsc[0])
__global__ __launch_bounds__(256) void split_copy_h_kernel(const float* __restrict__ x, float* __restrict__ xs, size_t n8,
                                                           const float* __restrict__ wsc) {
  const float sc = wsc[0];
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    float v[8];
    *reinterpret_cast<f32x4*>(v) = *reinterpret_cast<const f32x4*>(x + i * 8);
    *reinterpret_cast<f32x4*>(v + 4) = *reinterpret_cast<const f32x4*>(x + i * 8 + 4);
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] *= sc;
    split8h_store(v, xs + i * 8);
  }
}
// wsc[2] (float bits) = max over the rows of sum_k |hi(w[row][k] * wsc[0])|: the growth bound of the walk's per-token
// scaling; one block per row
__global__ __launch_bounds__(256) void rowabs_max_kernel(const float* __restrict__ w, int K, float* __restrict__ wsc) {
  __shared__ float red[256];
  const float* r = w + (size_t)blockIdx.x * K;
  const float sc = wsc[0];
  float s = 0.f;
  for (int k = threadIdx.x; k < K; k += 256) s += fabsf((float)(_Float16)(r[k] * sc));
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicMax(reinterpret_cast<unsigned*>(wsc + 2), __float_as_uint(red[0] * 1.0001f));
}
// fp32 activations -> fp16 pairs [hi8 | lo8] scaled by 2^k, k = floor(log2(30000 / max|x|)) from the ACT_MAX_SLOTS slots the
// producing conv raised (pooling keeps the maximum); block 0 also leaves 2^-k in *unscale for the consuming conv's epilogue
// (unscale also takes out the weight matrix' scale, wsc[1])
__global__ __launch_bounds__(256) void split_h_scaled_kernel(const float* __restrict__ x, float* __restrict__ xs, size_t n8,
                                                             const unsigned* __restrict__ max_slots, float* __restrict__ unscale,
                                                             const float* __restrict__ wsc) {
  float m = 0.f;
  for (int i = 0; i < ACT_MAX_SLOTS; ++i) m = fmaxf(m, __uint_as_float(max_slots[i]));
  int k = 0;
  if (m > 0.f && m < 3.0e38f) {
    k = (int)floorf(log2f(30000.f / m));
    k = k < -120 ? -120 : k > 120 ? 120 : k;
  }
  const float fac = ldexpf(1.f, k);
  if (blockIdx.x == 0 && threadIdx.x == 0) *unscale = ldexpf(1.f, -k) * wsc[1];
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    float v[8];
    *reinterpret_cast<f32x4*>(v) = *reinterpret_cast<const f32x4*>(x + i * 8);
    *reinterpret_cast<f32x4*>(v + 4) = *reinterpret_cast<const f32x4*>(x + i * 8 + 4);
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] *= fac;
    split8h_store(v, xs + i * 8);
  }
}
// ---- pairs straight from the producer (no split pass between two convs of the fp16-pair forward) -------------------
// The scale of a layer's OUTPUT pairs has to be known before its conv runs, so it comes from a bound instead of the
// measured maximum: |a_l| <= max|x_l| * (largest absolute row sum of w_l) + max|b_l|, with max|x_l| the maximum the
// producer of x_l measured (exact).  The bound overshoots the true maximum by the factor a row sum overshoots a typical
// dot product (20 ... 100 for VGG-sized layers) — and since every layer starts again from a MEASURED maximum the overshoot
// does not compound.  What it costs: a pair keeps 22 bits down to |value| = 2^-3 of the scaled range's floor; below that
// its absolute error stays 2^-25 of the SCALED unit, i.e. (overshoot / 30000) * 2^-25 relative to the tensor's maximum
// — 1e-10 at an overshoot of 100, far below the pair's own 2^-22.
// norm = { largest row sum of |w| over the output channels, max|b| } of the layer (conv_norm_kernel).
// Per IMAGE (one block each): out_scale[n] <- 2^k, k = floor(log2(30000 / bound_n)), bound_n from the maximum image n's rows
// raised in in_slots[n][ACT_MAX_SLOTS];  unscale_next[n] <- 2^-k * (scale record of the CONSUMER's weights)[1].
__global__ __launch_bounds__(64) void fwd_scale_kernel(const unsigned* __restrict__ in_slots, const float* __restrict__ norm,
                                                       const float* __restrict__ wsc_next, float* __restrict__ out_scale,
                                                       float* __restrict__ unscale_next) {
  const int n = blockIdx.x;
  float m = __uint_as_float(in_slots[(size_t)n * ACT_MAX_SLOTS + (threadIdx.x & (ACT_MAX_SLOTS - 1))]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if (threadIdx.x != 0) return;
  const float bound = m * norm[0] + norm[1];
  int k = 0;
  if (bound > 0.f && bound < 3.0e38f) {
    k = (int)floorf(log2f(30000.f / bound));
    k = k < -120 ? -120 : k > 120 ? 120 : k;
  }
  out_scale[n] = ldexpf(1.f, k);
  if (unscale_next) unscale_next[n] = ldexpf(1.f, -k) * wsc_next[1];
}
// max|x| per image into slots[n][ACT_MAX_SLOTS]; grid (blocks per image, images)
__global__ __launch_bounds__(256) void absmax_img_slots_kernel(const f32x4* __restrict__ x, size_t per_img4, unsigned* __restrict__ slots) {
  const f32x4* xi = x + (size_t)blockIdx.y * per_img4;
  float m = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < per_img4; i += (size_t)gridDim.x * 256) {
    const f32x4 v = xi[i];
    m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0 && m > 0.f)
    atomicMax(slots + (size_t)blockIdx.y * ACT_MAX_SLOTS + ((blockIdx.x + (threadIdx.x >> 6)) & (ACT_MAX_SLOTS - 1)), __float_as_uint(m));
}
// norm[0] = max over rows of sum_k |w[row][k]| (blocks 0 .. rows-1), norm[1] = max|b| (block `rows`); norm zeroed before
__global__ __launch_bounds__(256) void conv_norm_kernel(const float* __restrict__ w, int rows, int K, const float* __restrict__ b,
                                                        int nb, float* __restrict__ norm) {
  __shared__ float red[256];
  float s = 0.f;
  if ((int)blockIdx.x < rows) {
    const float* r = w + (size_t)blockIdx.x * K;
    for (int k = threadIdx.x; k < K; k += 256) s += fabsf(r[k]);
  } else {
    for (int k = threadIdx.x; k < nb; k += 256) s = fmaxf(s, fabsf(b[k]));
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] = (int)blockIdx.x < rows ? red[threadIdx.x] + red[threadIdx.x + o] : fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0)
    atomicMax(reinterpret_cast<unsigned*>(norm + ((int)blockIdx.x < rows ? 0 : 1)), __float_as_uint(red[0] * ((int)blockIdx.x < rows ? 1.0001f : 1.f)));
}
// stacked dual matrix [a rows (C) | Z rows (C)][K] -> rows interleaved per 32 channels ([w | w+] of the same channels side by side)
__global__ __launch_bounds__(256) void dual_interleave_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int K) {
  const size_t total = (size_t)2 * C * K;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int row = (int)(i / K), k = (int)(i % K);
    const int half = row >= C ? 1 : 0, c = row - half * C;
    dst[(size_t)(64 * (c / 32) + 32 * half + (c & 31)) * K + k] = src[i];
  }
}
// 2x2/2 max-pool + first-arg-max gate (pool_gate_kernel) + the pooled activations as the next conv's operand (fp16 pairs
// scaled by scale[image]) in ONE pass over (a_l, Z+_l): replaces maxpool2_kernel + pool_gate_kernel + split_h_scaled_kernel.
// 8 channels per thread; a may alias g; xnext (fp32 pooled activations: the fine-tune step's kept input) may be null.
// gc / gpos (may be null): the same gate in COMPACT form — per window and channel its one non-zero value and that value's
// position p = 2 dy + dx — for the consumer of the compact pool interface (conv_igemm.h ConvArgs::up2_gc).
__global__ __launch_bounds__(256) void pool_gate_split_kernel(const float* a, const float* __restrict__ z, float* g,
                                                              float* __restrict__ pairs, float* __restrict__ xnext,
                                                              const float* __restrict__ scale, int NB, int H, int W, int C,
                                                              float* __restrict__ gc, unsigned char* __restrict__ gpos) {
  const int C8 = C >> 3, Ho = H >> 1, Wo = W >> 1;
  const size_t total = (size_t)NB * Ho * Wo * C8;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    size_t r = i / C8;
    const int wo = (int)(r % Wo);
    r /= Wo;
    const int ho = (int)(r % Ho);
    const int n = (int)(r / Ho);
    float av[4][8], zv[4][8];
    size_t off[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      off[p] = ((((size_t)n * H + 2 * ho + (p >> 1)) * W + 2 * wo + (p & 1)) * C) + 8 * c8;
      *reinterpret_cast<f32x4*>(av[p]) = *reinterpret_cast<const f32x4*>(a + off[p]);
      *reinterpret_cast<f32x4*>(av[p] + 4) = *reinterpret_cast<const f32x4*>(a + off[p] + 4);
      *reinterpret_cast<f32x4*>(zv[p]) = *reinterpret_cast<const f32x4*>(z + off[p]);
      *reinterpret_cast<f32x4*>(zv[p] + 4) = *reinterpret_cast<const f32x4*>(z + off[p] + 4);
    }
    float mx[8];
    int arg[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { mx[c] = av[0][c]; arg[c] = 0; }
#pragma unroll
    for (int p = 1; p < 4; ++p)
#pragma unroll
      for (int c = 0; c < 8; ++c)
        if (av[p][c] > mx[c]) { mx[c] = av[p][c]; arg[c] = p; }
    const size_t po = (((size_t)n * Ho + ho) * Wo + wo) * C + 8 * c8;
    if (xnext) {
      *reinterpret_cast<f32x4*>(xnext + po) = *reinterpret_cast<const f32x4*>(mx);
      *reinterpret_cast<f32x4*>(xnext + po + 4) = *reinterpret_cast<const f32x4*>(mx + 4);
    }
    float sv[8];
    const float sc = scale[n];                          // per image (fwd_scale_kernel)
#pragma unroll
    for (int c = 0; c < 8; ++c) sv[c] = mx[c] * sc;
    split8h_store(sv, pairs + po);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      float o[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) o[c] = (arg[c] == p) ? av[p][c] / safe_den(zv[p][c]) : 0.f;
      *reinterpret_cast<f32x4*>(g + off[p]) = *reinterpret_cast<const f32x4*>(o);
      *reinterpret_cast<f32x4*>(g + off[p] + 4) = *reinterpret_cast<const f32x4*>(o + 4);
    }
    if (gc) {
      float o[8];
      unsigned pw[2] = {0u, 0u};
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        float av_ = av[0][c], zv_ = zv[0][c];
#pragma unroll
        for (int p = 1; p < 4; ++p)
          if (arg[c] == p) { av_ = av[p][c]; zv_ = zv[p][c]; }
        o[c] = av_ / safe_den(zv_);                       // (the very division of the full-resolution gate above)
        pw[c >> 2] |= (unsigned)arg[c] << (8 * (c & 3));
      }
      *reinterpret_cast<f32x4*>(gc + po) = *reinterpret_cast<const f32x4*>(o);
      *reinterpret_cast<f32x4*>(gc + po + 4) = *reinterpret_cast<const f32x4*>(o + 4);
      typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
      u32x2_ pv; pv[0] = pw[0]; pv[1] = pw[1];
      *reinterpret_cast<u32x2_*>(gpos + po) = pv;
    }
  }
}
// the full-resolution pool gate from its compact form (one value per window and channel + its position byte): what the fused
// pool epilogue of the forward conv (conv_igemm.h ConvArgs::pool_gc) leaves for the walks that read the expanded interface —
// the gradient baselines, the fp32 / fast modes, LRP_UP2_COMPACT=0 — built on their first use after an encode.  8 channels per thread.
__global__ __launch_bounds__(256) void pool_gate_expand_kernel(const float* __restrict__ gc, const unsigned char* __restrict__ gpos,
                                                               float* __restrict__ g, int NB, int H, int W, int C) {
  const int C8 = C >> 3, Ho = H >> 1, Wo = W >> 1;
  const size_t total = (size_t)NB * Ho * Wo * C8;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    size_t r = i / C8;
    const int wo = (int)(r % Wo);
    r /= Wo;
    const int ho = (int)(r % Ho);
    const int n = (int)(r / Ho);
    const size_t po = (((size_t)n * Ho + ho) * Wo + wo) * C + 8 * c8;
    float v[8];
    *reinterpret_cast<f32x4*>(v) = *reinterpret_cast<const f32x4*>(gc + po);
    *reinterpret_cast<f32x4*>(v + 4) = *reinterpret_cast<const f32x4*>(gc + po + 4);
    typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
    const u32x2_ pv = *reinterpret_cast<const u32x2_*>(gpos + po);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      float o[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) o[c] = ((pv[c >> 2] >> (8 * (c & 3))) & 0xFFu) == (unsigned)p ? v[c] : 0.f;
      float* dst = g + ((((size_t)n * H + 2 * ho + (p >> 1)) * W + 2 * wo + (p & 1)) * C) + 8 * c8;
      *reinterpret_cast<f32x4*>(dst) = *reinterpret_cast<const f32x4*>(o);
      *reinterpret_cast<f32x4*>(dst + 4) = *reinterpret_cast<const f32x4*>(o + 4);
    }
  }
}
// max|x| of a tensor into ACT_MAX_SLOTS slots (the image layer's activations come from the fp32 kernel, which keeps no maximum)
__global__ __launch_bounds__(256) void absmax_slots_kernel(const f32x4* __restrict__ x, size_t n4, unsigned* __restrict__ slots) {
  float m = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const f32x4 v = x[i];
    m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(slots + ((blockIdx.x + (threadIdx.x >> 6)) & (ACT_MAX_SLOTS - 1)), __float_as_uint(m));
}
// Between two launches of the PREC_F16X2 walk: the power of two the next conv adds to each token's scale (ConvArgs::tok_fac)
//   k = floor(log2(30000 / (max_in * wsc[2]))),  fac = 2^k,  exp_out = exp_in + k + kw     (wsc: the layer's weight record,
//   wsc[2] = norm of the scaled weights, kw = wsc[3] their scale exponent)
// final = 1 (in front of the image layer, whose output is plain fp32): fac = 2^-(exp_in + kw), nothing else.
__global__ __launch_bounds__(256) void tok_scale_kernel(const unsigned* __restrict__ max_in, const int* __restrict__ exp_in,
                                                        const float* __restrict__ wsc, float* __restrict__ fac,
                                                        int* __restrict__ exp_out, int n, int final) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  const int kw = (int)wsc[3];
  if (final) { fac[t] = ldexpf(1.f, -(exp_in[t] + kw)); return; }
  const float m = __uint_as_float(max_in[t]);
  int k = 0;
  if (m > 0.f) {
    k = (int)floorf(log2f(30000.f / (m * wsc[2])));
    k = k < -120 ? -120 : k > 120 ? 120 : k;
  }
  fac[t] = ldexpf(1.f, k);
  exp_out[t] = exp_in[t] + k + kw;
}
// Head of the reverse walk in PREC_F16X2: S_top = R_feat / safe(Z_top[img]) per token (KG:898-900), stored as fp16 pairs
// scaled by 2^k, k = floor(log2(30000 / max|S_top|)) — one workgroup per token, two passes over its 0.4 MB.
__global__ __launch_bounds__(256) void top_divide_f16_kernel(const float* __restrict__ R, const float* __restrict__ Ztop,
                                                             const int* __restrict__ row2img, float* __restrict__ S,
                                                             size_t per_img8, int* __restrict__ tok_exp, unsigned* __restrict__ tok_max) {
  __shared__ float red[256];
  const int t = blockIdx.x, img = row2img ? row2img[t] : t;
  const float* r = R + (size_t)t * per_img8 * 8;
  const float* z = Ztop + (size_t)img * per_img8 * 8;
  float m = 0.f;
  for (size_t i = threadIdx.x; i < per_img8; i += 256) {
    const f32x4 r0 = *reinterpret_cast<const f32x4*>(r + i * 8), r1 = *reinterpret_cast<const f32x4*>(r + i * 8 + 4);
    const f32x4 z0 = *reinterpret_cast<const f32x4*>(z + i * 8), z1 = *reinterpret_cast<const f32x4*>(z + i * 8 + 4);
#pragma unroll
    for (int c = 0; c < 4; ++c) m = fmaxf(m, fmaxf(fabsf(r0[c] / safe_den(z0[c])), fabsf(r1[c] / safe_den(z1[c]))));
  }
  red[threadIdx.x] = m;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
    __syncthreads();
  }
  m = red[0];
  int k = 0;
  if (m > 0.f && m < 3.0e38f) {
    k = (int)floorf(log2f(30000.f / m));
    k = k < -120 ? -120 : k > 120 ? 120 : k;
  }
  const float fac = ldexpf(1.f, k);
  if (threadIdx.x == 0) {
    tok_exp[t] = k;
    tok_max[t] = __float_as_uint(m * fac);
  }
  float* s = S + (size_t)t * per_img8 * 8;
  for (size_t i = threadIdx.x; i < per_img8; i += 256) {
    const f32x4 r0 = *reinterpret_cast<const f32x4*>(r + i * 8), r1 = *reinterpret_cast<const f32x4*>(r + i * 8 + 4);
    const f32x4 z0 = *reinterpret_cast<const f32x4*>(z + i * 8), z1 = *reinterpret_cast<const f32x4*>(z + i * 8 + 4);
    float v[8];
#pragma unroll
    for (int c = 0; c < 4; ++c) { v[c] = r0[c] / safe_den(z0[c]) * fac; v[4 + c] = r1[c] / safe_den(z1[c]) * fac; }
    split8h_store(v, s + i * 8);
  }
}

// Device twin of the image-layer packers of Encoder::set_conv_weight (li == 0): w (3,3,3,cout) HWIO ->
//   fwd  [.][64]  rows [0,cout): w against both halves of the im2col row (x+ patch | x- patch) = a_1;
//                 rows [cout,2cout): w+ | w- = Z_1 (RR:256-260)
//   bwd  [.][Kb]  row t*6+c = w+[t][c][:], row t*6+3+c = w-[t][c][:]   (tap-expanded channel reduction, conv_igemm.h)
//   full [.][Kb]  row t*6+c = w[t][c][:]                               (gradient baselines)
// One thread per (k = tap*3 + c, co); the buffers were zeroed at allocation, padding is never written.
__global__ __launch_bounds__(256) void pack_image_layer_dev_kernel(const float* __restrict__ w, float* __restrict__ fwd,
                                                                   float* __restrict__ bwd, float* __restrict__ full, int cout, int Kb) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 27 * cout) return;
  const int k = i / cout, co = i - k * cout, t = k / 3, c = k - 3 * t;
  const float v = w[i], vp = v >= 0.f ? v : 0.f, vn = v < 0.f ? v : 0.f;
  fwd[(size_t)co * 64 + k] = v;
  fwd[(size_t)co * 64 + 32 + k] = v;
  fwd[(size_t)(cout + co) * 64 + k] = vp;
  fwd[(size_t)(cout + co) * 64 + 32 + k] = vn;
  bwd[(size_t)(t * 6 + c) * Kb + co] = vp;
  bwd[(size_t)(t * 6 + 3 + c) * Kb + co] = vn;
  full[(size_t)(t * 6 + c) * Kb + co] = v;
}
// split8-packed [64][K] -> fragment-major copy for the weights-in-registers kernel (pack_frag64)
__global__ __launch_bounds__(256) void pack_frag64_dev_kernel(const float* __restrict__ src, float* __restrict__ dst, int CP) {
  const int K = 9 * CP, cpt = CP / 32;
  const size_t total = (size_t)cpt * 9 * 4 * 2 * 64;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int n = (int)(i & 63), hh = (int)((i >> 6) & 1), q = (int)((i >> 7) & 3);
    const int kc = (int)(i >> 9), cc = kc / 9, t = kc % 9;
    const int c = 4 * (q >> 1) + 2 * hh + (q & 1);
    *reinterpret_cast<f32x4*>(dst + i * 4) = *reinterpret_cast<const f32x4*>(src + (size_t)n * K + t * CP + cc * 32 + c * 4);
  }
}

// Head of the reverse walk (KG:898-900): S_top = R_feat / safe(Z_top[img])
__global__ __launch_bounds__(256) void top_divide_kernel(const f32x4* __restrict__ R, const f32x4* __restrict__ Ztop,
                                                         const int* __restrict__ row2img, f32x4* __restrict__ S,
                                                         int n, size_t per_img4) {
  const size_t total = (size_t)n * per_img4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int t = (int)(i / per_img4);
    const size_t e = i - (size_t)t * per_img4;
    const int img = row2img ? row2img[t] : t;
    const f32x4 r = R[i], z = Ztop[(size_t)img * per_img4 + e];
    f32x4 o;
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = r[c] / safe_den(z[c]);
    S[i] = o;
  }
}

// Same head, writing the split8 operand format of the bf16x3 reverse walk (8 channels per thread)
__global__ __launch_bounds__(256) void top_divide_split_kernel(const float* __restrict__ R, const float* __restrict__ Ztop,
                                                               const int* __restrict__ row2img, float* __restrict__ S,
                                                               int n, size_t per_img8) {
  const size_t total = (size_t)n * per_img8;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int t = (int)(i / per_img8);
    const size_t e = i - (size_t)t * per_img8;
    const int img = row2img ? row2img[t] : t;
    const float* r = R + i * 8;
    const float* z = Ztop + ((size_t)img * per_img8 + e) * 8;
    float o[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) o[c] = r[c] / safe_den(z[c]);
    split8_store(o, S + i * 8);
  }
}

// Image layer of the reverse walk (RR:306-312 with both sign branches live):
//   R_img[p][c] = x+[p][c] * convT(S_1, w+)[p][c] + x-[p][c] * convT(S_1, w-)[p][c]
// computed as a channel reduction FIRST (one K = C_1 GEMM on the MFMA kernel):
//   T[q][tap*6 + c]   = sum_co S_1[q][co] * w+[tap][c][co]      (c = 0..2)
//   T[q][tap*6 + 3+c] = sum_co S_1[q][co] * w-[tap][c][co]
// and then this 9-tap shift-and-add:  convT(S,w)[p] = sum_tap T[p - d(tap)][tap],  d = (kh-1, kw-1).
// (A direct N = 6 implicit GEMM would pad N to 32 and burn 5x the MFMA work.)
constexpr int IMG_T_COLS = 54;
__global__ __launch_bounds__(256) void img_stencil_kernel(const float* __restrict__ T, const float* __restrict__ ximg,
                                                          const int* __restrict__ row2img, float* __restrict__ out,
                                                          int n, int H, int W, int mode) {
  const size_t total = (size_t)n * H * W;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int HW = H * W;
    const int t = (int)(i / HW), pix = (int)(i - (size_t)t * HW);
    const int h = pix / W, w = pix - h * W;
    float pos[3] = {0.f, 0.f, 0.f}, neg[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int hh = h - (tap / 3 - 1), ww = w - (tap % 3 - 1);
      if (hh >= 0 && hh < H && ww >= 0 && ww < W) {
        const float* r = T + (((size_t)t * H + hh) * W + ww) * IMG_T_COLS + tap * 6;
        const float2 a = *reinterpret_cast<const float2*>(r);
        const float2 b = *reinterpret_cast<const float2*>(r + 2);
        const float2 c = *reinterpret_cast<const float2*>(r + 4);
        pos[0] += a.x; pos[1] += a.y; pos[2] += b.x;
        neg[0] += b.y; neg[1] += c.x; neg[2] += c.y;
      }
    }
    const int img = row2img ? row2img[t] : t;
    const float* x = ximg + ((size_t)img * HW + pix) * 3;
    float* o = out + i * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      if (mode == 0) o[c] = x[c] >= 0.f ? x[c] * pos[c] : x[c] * neg[c];   // LRP alpha1beta0 at the image
      else if (mode == 1) o[c] = pos[c] + neg[c];                          // plain gradient (w sits in the + columns)
      else o[c] = x[c] * (pos[c] + neg[c]);                                // input x gradient
    }
  }
}

// Second half of the image layer when it is folded into the epilogue of the layer above it (conv_igemm.h ConvArgs::img_part):
// every tile of that launch left, for its (th + 2) x (tw + 2) ring-inclusive output positions, the six partial sums
// (3 x "+", 3 x "-") over the source pixels it owns.  An output pixel adds the partials of the up to four tiles whose ring
// covers it — in a FIXED order (tile row, then tile column), the same for every token and every batch — and applies
//   R_img = x+ * sum+  +  x- * sum-      (RR:274-322 at the image; mode 1 / 2: plain gradient / input x gradient).
__global__ __launch_bounds__(256) void img_partial_sum_kernel(const float* __restrict__ part, const float* __restrict__ ximg,
                                                              const int* __restrict__ row2img, float* __restrict__ out, int n, int H, int W,
                                                              int th, int tw, int cols_t, int mode) {
  const size_t total = (size_t)n * H * W;
  const int RW = tw + 2, npos = (th + 2) * RW, tpt = (H + th - 1) / th;   // tiles are laid out per token: tpt tile rows each
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int HW = H * W;
    const int t = (int)(i / HW), pix = (int)(i - (size_t)t * HW);
    const int h = pix / W, w = pix - h * W;
    const int tr0 = h / th, txt0 = w / tw, hm = h - tr0 * th, wm = w - txt0 * tw;        // tile row WITHIN the token
    const int ty_lo = (hm == 0 && h > 0) ? tr0 - 1 : tr0, ty_hi = (hm == th - 1 && h + 1 < H) ? tr0 + 1 : tr0;
    const int tx_lo = (wm == 0 && w > 0) ? txt0 - 1 : txt0, tx_hi = (wm == tw - 1 && txt0 + 1 < cols_t) ? txt0 + 1 : txt0;
    float s[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int ty = ty_lo; ty <= ty_hi; ++ty)
      for (int tx = tx_lo; tx <= tx_hi; ++tx) {
        const int p = (h - ty * th + 1) * RW + (w - tx * tw + 1);
        const float* r = part + ((((size_t)t * tpt + ty) * cols_t + tx) * npos + p) * 6;
        const float2 a = *reinterpret_cast<const float2*>(r);
        const float2 b = *reinterpret_cast<const float2*>(r + 2);
        const float2 c = *reinterpret_cast<const float2*>(r + 4);
        s[0] += a.x; s[1] += a.y; s[2] += b.x; s[3] += b.y; s[4] += c.x; s[5] += c.y;
      }
    const int img = row2img ? row2img[t] : t;
    const float* x = ximg + ((size_t)img * HW + pix) * 3;
    float* o = out + i * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      if (mode == 0) o[c] = x[c] >= 0.f ? x[c] * s[c] : x[c] * s[3 + c];
      else if (mode == 1) o[c] = s[c] + s[3 + c];
      else o[c] = x[c] * (s[c] + s[3 + c]);
    }
  }
}

// Head of the gradient walks: the cut is AFTER block5_conv3's ReLU, so the head tensor first passes that ReLU's
// backward: S_top = R * [feat > 0]  (guided backprop: max(R, 0) * [feat > 0], gradient_based.py:228-234)
__global__ __launch_bounds__(256) void grad_top_kernel(const f32x4* __restrict__ R, const f32x4* __restrict__ feat,
                                                       const int* __restrict__ row2img, f32x4* __restrict__ S, int n,
                                                       size_t per4, int guided) {
  const size_t total = (size_t)n * per4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t t = i / per4, r = i - t * per4;
    const int img = row2img ? row2img[t] : (int)t;
    const f32x4 rv = R[i], fv = feat[(size_t)img * per4 + r];
    f32x4 o;
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = fv[c] > 0.f ? (guided ? fmaxf(rv[c], 0.f) : rv[c]) : 0.f;
    S[i] = o;
  }
}

inline int stream_grid(size_t work_items) {
  size_t b = (work_items + 255) / 256;
  if (b > 256 * 8) b = 256 * 8;      // 8 blocks per CU, grid-stride the rest
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace lrp
