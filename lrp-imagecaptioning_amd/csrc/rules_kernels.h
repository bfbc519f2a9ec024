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

}  // namespace lrp
