// score_kernels.h — heat-map -> scalar score of the LRP-inference layer (models/model.py:1675-1686),
// one workgroup per heat-map, HBM-bound (each pass streams the 600 KB map from L2/HBM):
//   hp = mean_c(R) ;  hp /= max|hp| (0 if the map is all zero) ;
//   mode 0 "mean": mean(hp)   mode 1 "pos_mean": mean(max(hp,0))   mode 2 "quantile": np.quantile(hp, 0.9)
// (postprocess()'s BGR->RGB flip does not change a channel mean.)  The quantile is exact: a 3-pass
// 11/11/10-bit radix select on the order-preserving integer image of the float keys finds the two order
// statistics numpy interpolates between (method 'linear': index (n-1)*0.9).
#pragma once
#include <hip/hip_runtime.h>
#include "decoder_kernels.h"

namespace lrp {

__device__ __forceinline__ unsigned f2key(float f) {           // monotone float -> uint
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
  const unsigned u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
  return __uint_as_float(u);
}
__device__ __forceinline__ float hp_at(const float* __restrict__ R, size_t p, int C) {
  // np.mean(hp, axis=-1) on float32: sequential float32 adds, then / C
  float s = 0.f;
  for (int c = 0; c < C; ++c) s += R[p * C + c];
  return s / (float)C;
}

// k-th smallest (0-based) of hp over the map, exact, via radix select; all threads of the block call it
__device__ float select_kth(const float* __restrict__ R, int npix, int C, unsigned k, unsigned* hist /*2048*/,
                            unsigned* bcast /*2*/) {
  const int tid = threadIdx.x;
  unsigned prefix = 0, mask = 0;
  const int shifts[3] = {21, 10, 0}, bits[3] = {11, 11, 10};
  for (int pass = 0; pass < 3; ++pass) {
    const int nb = 1 << bits[pass];
    for (int i = tid; i < 2048; i += 256) hist[i] = 0;
    __syncthreads();
    for (int p = tid; p < npix; p += 256) {
      const unsigned key = f2key(hp_at(R, p, C));
      if ((key & mask) == prefix) atomicAdd(&hist[(key >> shifts[pass]) & (nb - 1)], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      unsigned acc = 0;
      int b = 0;
      for (; b < nb; ++b) {
        if (acc + hist[b] > k) break;
        acc += hist[b];
      }
      bcast[0] = (unsigned)b;
      bcast[1] = acc;
    }
    __syncthreads();
    prefix |= bcast[0] << shifts[pass];
    mask |= (unsigned)(nb - 1) << shifts[pass];
    k -= bcast[1];
    __syncthreads();
  }
  return key2f(prefix);
}

__global__ __launch_bounds__(256) void heatmap_score_kernel(const float* __restrict__ Rall, double* __restrict__ scores,
                                                            int npix, int C, int mode) {
  __shared__ unsigned hist[2048];
  __shared__ unsigned bcast[2];
  __shared__ double red[4];
  __shared__ float fred[4];
  const float* R = Rall + (size_t)blockIdx.x * npix * C;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // pass 1: max |hp|
  float mx = 0.f;
  for (int p = tid; p < npix; p += 256) mx = fmaxf(mx, fabsf(hp_at(R, p, C)));
  mx = wave_max(mx);
  if (lane == 0) fred[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(fred[0], fred[1]), fmaxf(fred[2], fred[3]));
  __syncthreads();
  if (mx == 0.f) {                                   // project(): all-zero map -> zeros
    if (tid == 0) scores[blockIdx.x] = 0.0;
    return;
  }
  if (mode == 2) {
    // np.quantile(hp/mx, 0.9), linear interpolation between the two neighbouring order statistics
    const double pos = (double)(npix - 1) * 0.9;
    const unsigned k0 = (unsigned)pos;
    const double frac = pos - (double)k0;
    const float v0 = select_kth(R, npix, C, k0, hist, bcast);
    const float v1 = (k0 + 1 < (unsigned)npix) ? select_kth(R, npix, C, k0 + 1, hist, bcast) : v0;
    if (tid == 0) {
      const double a = (double)(v0 / mx), b = (double)(v1 / mx);      // hp = 1.0 * hp / absmax in float32
      scores[blockIdx.x] = a + (b - a) * frac;
    }
    return;
  }
  double s = 0.0;
  for (int p = tid; p < npix; p += 256) {
    const float v = hp_at(R, p, C) / mx;
    s += (double)(mode == 1 ? fmaxf(v, 0.f) : v);
  }
  s = block_sum_d(s, red);
  if (tid == 0) scores[blockIdx.x] = s / (double)npix;
}

// Heat-map rendering of the explanation harness (innvestigate/examples/utils_imagenet.py:31-33 ->
// utils/visualizations.py:87-125, :57-80, :36-54): per heat-map
//   Y = sign(x) (|x| / max|x|)^gamma max|x|   (gamma = 0.95, float32 like numpy on a float32 array)
//   t = sum_c Y;  idx = int( clip((t / max|t| + 1) / 2, 0, 1) * 255 )   (float64, truncation like astype(int))
//   rgb = lut[idx]                              (the 256-entry colormap table, 'seismic' in the reference)
// One workgroup per heat-map, three passes over its 600 KB (L2 resident after the first).
__global__ __launch_bounds__(256) void heatmap_render_kernel(const float* __restrict__ Rall, const float* __restrict__ lut,
                                                             float* __restrict__ out, int npix, int C, float gam) {
  __shared__ float fred[4];
  __shared__ float bc;
  const float* R = Rall + (size_t)blockIdx.x * npix * C;
  float* o = out + (size_t)blockIdx.x * npix * 3;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  auto block_max = [&](float v) {
    v = wave_max(v);
    __syncthreads();
    if (lane == 0) fred[wave] = v;
    __syncthreads();
    if (tid == 0) bc = fmaxf(fmaxf(fred[0], fred[1]), fmaxf(fred[2], fred[3]));
    __syncthreads();
    return bc;
  };
  float mx = 0.f;
  for (int i = tid; i < npix * C; i += 256) mx = fmaxf(mx, fabsf(R[i]));
  const float maxamp = block_max(mx);
  auto tsum = [&](int p) {
    float t = 0.f;
    for (int c = 0; c < C; ++c) {
      const float x = R[(size_t)p * C + c] / maxamp;
      const float y = x >= 0.f ? powf(x, gam) : -powf(-x, gam);
      t += y * maxamp;
    }
    return t;
  };
  float mt = 0.f;
  if (maxamp > 0.f)
    for (int p = tid; p < npix; p += 256) mt = fmaxf(mt, fabsf(tsum(p)));
  const float absmax = block_max(mt);
  for (int p = tid; p < npix; p += 256) {
    double v = maxamp > 0.f ? (double)tsum(p) : 0.0;
    if (absmax != 0.f) v /= (double)absmax;
    v = (v + 1.0) / 2.0;
    v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
    const int idx = (int)(v * 255.0);
    o[(size_t)p * 3] = lut[idx * 3]; o[(size_t)p * 3 + 1] = lut[idx * 3 + 1]; o[(size_t)p * 3 + 2] = lut[idx * 3 + 2];
  }
}

// Image preprocessing of models/preprocessors.py:38-53 for the caffe-style encoders (vgg16 / vgg19 / resnet101 all use
// keras `preprocess_input` in 'caffe' mode): decoded RGB bytes (H0, W0, 3) -> nearest-neighbour resize to (H, W) as
// PIL does for keras `load_img(target_size=...)` (source pixel = floor((o + 0.5) * H0 / H)), RGB -> BGR, minus the
// ImageNet channel means.  One launch for a batch of equally sized inputs.
__global__ __launch_bounds__(256) void preprocess_caffe_kernel(const unsigned char* __restrict__ rgb, float* __restrict__ out,
                                                              int NB, int H0, int W0, int H, int W) {
  const float mean[3] = {103.939f, 116.779f, 123.68f};                    // B, G, R
  const size_t total = (size_t)NB * H * W;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int x = (int)(i % W);
    size_t r = i / W;
    const int y = (int)(r % H);
    const int n = (int)(r / H);
    // PIL's ImagingScaleAffine walks the source coordinate incrementally in double (xo = a0/2; xo += a0 per output
    // pixel) and truncates; where (o + 0.5) * a0 is an exact integer the accumulated rounding decides the pixel, so the
    // recurrence is replayed literally (<= 224 additions per axis) rather than evaluated in closed form.
    const double fy = (double)H0 / (double)H, fx = (double)W0 / (double)W;
    double yo = fy * 0.5, xo = fx * 0.5;
    for (int q = 0; q < y; ++q) yo += fy;
    for (int q = 0; q < x; ++q) xo += fx;
    int sy = (int)yo, sx = (int)xo;
    sy = sy < H0 ? sy : H0 - 1;
    sx = sx < W0 ? sx : W0 - 1;
    const unsigned char* p = rgb + (((size_t)n * H0 + sy) * W0 + sx) * 3;
    float* o = out + i * 3;
    o[0] = (float)p[2] - mean[0];
    o[1] = (float)p[1] - mean[1];
    o[2] = (float)p[0] - mean[2];
  }
}

// ---- caption generation: log-soft-max + top-k of one row of un-normalised scores per workgroup, so that a
// beam-search step moves only k (id, log p) pairs per hypothesis over PCIe instead of the (beams, V) logits.
// Reference: `_log_softmax` E:45-48 followed by `np.argpartition(preds, -beam_size)[:, -beam_size:]` E:76-78
// (models/explainers.py) / inference.py:205-214.  logits (rows, V) float64 (what lrp_decoder_gen_step writes);
// ids (rows, k) int32 = model columns (tokenizer id - 1, E:92) in descending order of probability, ties to the
// lower column; logp (rows, k) float64 = x - max - log(sum(exp(x - max))).  HBM/L2-bound: k + 2 passes over 8 V bytes.
constexpr int TOPK_MAX = 32;
__global__ __launch_bounds__(256) void log_softmax_topk_kernel(const double* __restrict__ logits, int V, int k,
                                                                int* __restrict__ ids, double* __restrict__ logp) {
  __shared__ double red[256];
  __shared__ int redi[256];
  __shared__ int chosen[TOPK_MAX];
  const int tid = threadIdx.x;
  const double* x = logits + (size_t)blockIdx.x * V;
  double m = -1.0 / 0.0;
  for (int i = tid; i < V; i += 256) m = fmax(m, x[i]);
  red[tid] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] = fmax(red[tid], red[tid + s]);
    __syncthreads();
  }
  m = red[0];
  __syncthreads();
  double sum = 0.0;
  for (int i = tid; i < V; i += 256) sum += exp(x[i] - m);
  red[tid] = sum;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {                  // fixed tree: the same bits run to run
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  const double lse = log(red[0]);
  __syncthreads();
  for (int j = 0; j < k; ++j) {
    double best = -1.0 / 0.0;
    int bi = 0x7fffffff;
    for (int i = tid; i < V; i += 256) {
      bool taken = false;
      for (int q = 0; q < j; ++q) taken |= chosen[q] == i;
      const double v = x[i];
      if (!taken && (v > best || (v == best && i < bi))) { best = v; bi = i; }
    }
    red[tid] = best;
    redi[tid] = bi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (tid < s) {
        const double o = red[tid + s];
        const int oi = redi[tid + s];
        if (o > red[tid] || (o == red[tid] && oi < redi[tid])) { red[tid] = o; redi[tid] = oi; }
      }
      __syncthreads();
    }
    if (tid == 0) {
      chosen[j] = redi[0];
      ids[(size_t)blockIdx.x * k + j] = redi[0];
      logp[(size_t)blockIdx.x * k + j] = red[0] - m - lse;
    }
    __syncthreads();
  }
}

}  // namespace lrp
