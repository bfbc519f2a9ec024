// rules_kernels.h — the remaining LRP rules the preset can reach, as operator-level HIP
// kernels.  They are not exercised by the truncated VGG16 encoder
// (no Dense / BatchNorm / Add before block5_conv3); they exist for the ResNet-101 "next" row
// and are parity-tested on their own.  All streaming, HBM-bound.
#pragma once
#include <hip/hip_runtime.h>
#include "cnn_kernels.h"

namespace lrp {

// EpsilonRule (RR:113-144), PresetA passes bias=False (RA:706-711):
//   S = R / (Z + (2[Z>=0]-1) * eps)      — the GEMMs Z = x.W and C = S.W^T run on conv_igemm (taps = 1)
__global__ __launch_bounds__(256) void eps_divide_kernel(const float* __restrict__ R, const float* __restrict__ Z,
                                                         float* __restrict__ S, float eps, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float z = Z[i];
    S[i] = R[i] / (z + (z >= 0.f ? eps : -eps));
  }
}

// BatchNormalizationReverseLayer (RA:197-257), channels-last, inference statistics:
//   y = gamma (x - mean) / sqrt(var + bn_eps) + beta
//   R_in = SafeDivide( x (y - beta) R ,  stab((x - mean) y) ),  stab(d) = d + (2[d>=0]-1) 1e-7
__global__ __launch_bounds__(256) void bn_lrp_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, const float* __restrict__ mean,
                                                     const float* __restrict__ var, float bn_eps,
                                                     const float* __restrict__ R, float* __restrict__ out, size_t n, int C) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const float xv = x[i], mu = mean[c], bt = beta[c];
    const float y = gamma[c] * (xv - mu) / sqrtf(var[c] + bn_eps) + bt;
    const float num = xv * (y - bt) * R[i];
    float den = (xv - mu) * y;
    den += den >= 0.f ? 1e-7f : -1e-7f;
    out[i] = num / safe_den(den);
  }
}

// AddReverseLayer (RA:260-286): R_i = x_i * SafeDivide(R, x_a + x_b)
__global__ __launch_bounds__(256) void add_lrp_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      const float* __restrict__ R, float* __restrict__ Ra,
                                                      float* __restrict__ Rb, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float av = a[i], bv = b[i];
    const float s = R[i] / safe_den(av + bv);
    Ra[i] = av * s;
    Rb[i] = bv * s;
  }
}

// AveragePoolingReverseLayer (RA:289-316), k x k / stride k 'valid' average pooling on NHWC:
//   Z = avgpool(x);  S = SafeDivide(R, Z);  R_in = x * dZ/dx^T S = x * S[window(x)] / k^2
// one thread per 4 channels of one OUTPUT window: Z is a k*k-term sum it then fans back out (x read twice, from L2).
__global__ __launch_bounds__(256) void avgpool_lrp_kernel(const float* __restrict__ x, const float* __restrict__ R,
                                                          float* __restrict__ out, int NB, int H, int W, int C, int k) {
  const int C4 = C >> 2, Ho = H / k, Wo = W / k;
  const size_t total = (size_t)NB * Ho * Wo * C4;
  const float inv = 1.0f / (float)(k * k);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    size_t r = i / C4;
    const int wo = (int)(r % Wo);
    r /= Wo;
    const int ho = (int)(r % Ho);
    const int n = (int)(r / Ho);
    const float* xb = x + ((((size_t)n * H + (size_t)ho * k) * W + (size_t)wo * k) * C) + 4 * c4;
    float* ob = out + (xb - x);
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (int dy = 0; dy < k; ++dy)
      for (int dx = 0; dx < k; ++dx) z += *reinterpret_cast<const f32x4*>(xb + ((size_t)dy * W + dx) * C);
    const f32x4 rv = *reinterpret_cast<const f32x4*>(R + (((size_t)n * Ho + ho) * Wo + wo) * C + 4 * c4);
    f32x4 s;
#pragma unroll
    for (int c = 0; c < 4; ++c) s[c] = rv[c] / safe_den(z[c] * inv) * inv;
    for (int dy = 0; dy < k; ++dy)
      for (int dx = 0; dx < k; ++dx) {
        const size_t o = ((size_t)dy * W + dx) * C;
        *reinterpret_cast<f32x4*>(ob + o) = *reinterpret_cast<const f32x4*>(xb + o) * s;
      }
  }
}

}  // namespace lrp
