#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x16 __attribute__((ext_vector_type(16)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int SPARSE>
__global__ __launch_bounds__(256) void probe(float* out, int iters, unsigned seed) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a; bf16x16 b16; bf16x8 b8;
  unsigned s = seed + threadIdx.x * 2654435761u;
  for (int q = 0; q < 8; ++q) { s = s * 1664525u + 1013904223u; a[q] = (__bf16)((float)(s >> 8) / 16777216.f - 0.5f); }
  for (int q = 0; q < 16; ++q) { s = s * 1664525u + 1013904223u; b16[q] = (__bf16)((float)(s >> 8) / 16777216.f - 0.5f); }
  for (int q = 0; q < 8; ++q) b8[q] = b16[q];
  int idx = 0x4444;  // 2-bit indices
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (SPARSE) acc[i] = __builtin_amdgcn_smfmac_f32_32x32x32_bf16(a, b16, acc[i], idx, 0, 0);
      else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b8, acc[i], 0, 0, 0);
    }
  }
  float t = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) t += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = t;
}
int main() {
  float* out; hipMalloc(&out, 1024 * 256 * 4);
  const int iters = 20000, blocks = 1024;
  for (int sp = 0; sp < 2; ++sp) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (sp) hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1u);
      else hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 1u);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double instr = (double)blocks * 4 /*waves*/ * iters * 4;
      const double dense_flop_per = 2.0 * 32 * 32 * 16;
      printf("%s: %.3f ms, %.1f G instr/s, dense-equivalent %.0f TFLOP/s (%s)\n", sp ? "smfmac_f32_32x32x32_bf16" : "mfma_f32_32x32x16_bf16  ", ms,
             instr / ms / 1e6, instr * dense_flop_per * (sp ? 2 : 1) / ms / 1e9, sp ? "counting the 32 uncompressed k" : "16 k");
    }
  }
  return 0;
}
