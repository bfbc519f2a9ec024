// bf16 weight gradients of the fine-tune step (BASELINE config 5 "bf16"): dW[tap] = X_tap^T . dZ with both
// operands rounded to bf16 on the way into LDS, v_mfma_f32_32x32x16_bf16, fp32 accumulation, fp32 partials / result.
//
// The contraction index of a weight gradient is the PIXEL (K = B*H*W rows of two NHWC tensors), i.e. both operands lie
// k-major in memory, while the bf16 MFMA wants 8 consecutive k per lane.  The fp32 kernel (train_gemm.h) side-steps that
// with the one-scalar-per-lane operand of v_mfma_f32_32x32x2_f32; here the transpose is the LDS read itself:
// ds_read_b64_tr_b16 hands each lane 4 consecutive ROWS (k) of one COLUMN (channel) of a row-major [k][channel] image —
// two of them are one MFMA fragment — so the tiles are staged exactly as they lie in memory (coalesced 32 B per thread,
// fp32 -> bf16 in registers, 16 B LDS writes) and never transposed by hand.
//
// LDS image per operand and buffer: 32 rows (k) x 256 B (128 channels x bf16), 16 B chunk index XOR-swizzled with
// ((row & 3) << 2) | ((row >> 2) & 3): the four rows of a transposed-read block land on four different 64 B bank
// quarters (unswizzled they share one: 4-way conflicts), and the staging writes still cover whole rows.
// Block tile (64 TM) x (64 TN), 2 x 2 waves of (32 TM) x (32 TN), k-slab 32, register prefetch of the next slab.
// The K split / tap folding / deterministic second pass are train_gemm.h's (sgemm_reduce_kernel).
#pragma once
#include <hip/hip_runtime.h>

#include "conv_igemm.h"
#include "train_gemm.h"

namespace lrp {

typedef short wg_s16x4 __attribute__((ext_vector_type(4)));
typedef short wg_s16x8 __attribute__((ext_vector_type(8)));
constexpr int WG_BK = 32;
constexpr int WG_ROWB = 256;                             // bytes per LDS row (128 bf16 columns; narrower tiles use a prefix)

__device__ __forceinline__ int wg_off(int row, int ch) { return WG_ROWB * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

// 32 k-rows x BMN columns of one operand: thread -> (row, 8-column chunk); fp32 from memory into registers
template <int BMN>
__device__ __forceinline__ void wg_fetch(const float* __restrict__ P, long ld, int cols_total, long k_end, int mn0, long k0, int tid,
                                         bool vec, const SgemmArgs& a, bool is_a, float (&r)[BMN / 64][8]) {
#pragma unroll
  for (int q = 0; q < BMN / 64; ++q) {
    const int idx = tid + 256 * q;
    const int kr = idx / (BMN / 8), c8 = idx % (BMN / 8);
    const long k = k0 + kr;
    const int col = mn0 + 8 * c8;
    long src = k;
    bool ok = k < k_end;
    if (is_a && a.gather && ok) {
      const unsigned ku = (unsigned)k, rowi = ku / (unsigned)a.gW;
      const int x = (int)(ku - rowi * (unsigned)a.gW), y = (int)(rowi % (unsigned)a.gH);
      ok = (unsigned)(y + a.dy) < (unsigned)a.gH && (unsigned)(x + a.dx) < (unsigned)a.gW;
      src = k + (long)a.dy * a.gW + a.dx;
    }
    if (ok && vec && col + 7 < cols_total) {
      const float4 v0 = *reinterpret_cast<const float4*>(P + src * ld + col);
      const float4 v1 = *reinterpret_cast<const float4*>(P + src * ld + col + 4);
      r[q][0] = v0.x; r[q][1] = v0.y; r[q][2] = v0.z; r[q][3] = v0.w;
      r[q][4] = v1.x; r[q][5] = v1.y; r[q][6] = v1.z; r[q][7] = v1.w;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) r[q][j] = (ok && col + j < cols_total) ? P[src * ld + col + j] : 0.f;
    }
  }
}
template <int BMN>
__device__ __forceinline__ void wg_put(unsigned char* __restrict__ S, int tid, const float (&r)[BMN / 64][8]) {
#pragma unroll
  for (int q = 0; q < BMN / 64; ++q) {
    const int idx = tid + 256 * q;
    const int kr = idx / (BMN / 8), c8 = idx % (BMN / 8);
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (__bf16)r[q][j];
    *reinterpret_cast<u32x4*>(S + wg_off(kr, c8)) = __builtin_bit_cast(u32x4, v);
  }
}

// one MFMA operand fragment (32 columns x 16 k) of k-step s from a [k][column] image: two transposed 4-row reads
__device__ __forceinline__ bf16x8 wg_frag(const unsigned char* S, int chunk_base, int s, int lane) {
  typedef __attribute__((address_space(3))) wg_s16x4* lp_t;
  const int h = lane >> 5, g2 = (lane >> 4) & 1, q = (lane >> 2) & 3, p = lane & 3;
  const int r0 = 16 * s + 8 * h;
  const int ch = chunk_base + 2 * g2 + (p >> 1);
  const wg_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(S + wg_off(r0 + q, ch) + 8 * (p & 1)));
  const wg_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(S + wg_off(r0 + 4 + q, ch) + 8 * (p & 1)));
  wg_s16x8 f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
  f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
  return __builtin_bit_cast(bf16x8, f);
}

// C[M][N] (+ per tap) = sum_k A[k][m] B[k][n], A gathered by the tap when a.gather; partials to a.ws like sgemm_kernel
template <int TM, int TN>
__global__ __launch_bounds__(256) void wgrad_bf16_kernel(SgemmArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = 64 * TM, BN = 64 * TN;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2][2][WG_BK * WG_ROWB];      // [buffer][A | B]: 32 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int tap = a.taps > 1 ? (int)blockIdx.z / a.ksplit : 0, slice = a.taps > 1 ? (int)blockIdx.z % a.ksplit : (int)blockIdx.z;
  if (a.taps > 1) { a.dy = tap / 3 - 1; a.dx = tap % 3 - 1; }
  const long kb = (long)slice * a.kchunk;
  const long ke = kb + a.kchunk < a.K ? kb + a.kchunk : a.K;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float ra[TM][8], rb[TN][8];
  const long nst = (ke - kb + WG_BK - 1) / WG_BK;
  if (nst > 0) {
    wg_fetch<BM>(a.A, a.lda, a.M, ke, m0, kb, tid, a.vecA != 0, a, true, ra);
    wg_fetch<BN>(a.B, a.ldb, a.N, ke, n0, kb, tid, a.vecB != 0, a, false, rb);
    wg_put<BM>(smem[0][0], tid, ra);
    wg_put<BN>(smem[0][1], tid, rb);
  }
  __syncthreads();
  for (long s = 0; s < nst; ++s) {
    const int cur = (int)(s & 1);
    const bool more = s + 1 < nst;
    if (more) {
      wg_fetch<BM>(a.A, a.lda, a.M, ke, m0, kb + (s + 1) * WG_BK, tid, a.vecA != 0, a, true, ra);
      wg_fetch<BN>(a.B, a.ldb, a.N, ke, n0, kb + (s + 1) * WG_BK, tid, a.vecB != 0, a, false, rb);
    }
#pragma unroll
    for (int ks = 0; ks < WG_BK / 16; ++ks) {
      bf16x8 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = wg_frag(smem[cur][0], (wm * TM + i) * 4, ks, lane);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = wg_frag(smem[cur][1], (wn * TN + j) * 4, ks, lane);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if (more) {
      wg_put<BM>(smem[cur ^ 1][0], tid, ra);
      wg_put<BN>(smem[cur ^ 1][1], tid, rb);
    }
    __syncthreads();
  }
  const bool partial = a.ksplit > 1 || a.taps > 1;
  float* C = partial ? a.ws + (size_t)blockIdx.z * a.M * a.N : a.C;
  const long ldc = partial ? a.N : a.ldc;
  const bool accum = partial ? false : a.accumulate != 0;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (row >= a.M) continue;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = n0 + (wn * TN + j) * 32 + (lane & 31);
        if (col < a.N) {
          float* p = C + (long)row * ldc + col;
          *p = accum ? *p + acc[i][j][r] : acc[i][j][r];
        }
      }
    }
#endif
}

// C (+)= A^T B with bf16 operands (A: K x M, B: K x N, both k-major fp32 in memory); same contract as sgemm() for the
// TN form incl. `gather` / `taps`.  M, N tiles of 64 or 128; the K split fills the chip (~4 workgroups per CU).
inline hipError_t wgrad_bf16(SgemmArgs a, float* ws, size_t ws_floats, hipStream_t st) {
  if (a.M < 1 || a.N < 1 || a.K < 1 || !a.transA || a.transB) return hipErrorInvalidValue;
  a.vecA = sg_aligned(a.A, a.lda) ? 1 : 0;
  a.vecB = sg_aligned(a.B, a.ldb) ? 1 : 0;
  const int tm = a.M <= 64 ? 1 : 2, tn = a.N <= 64 ? 1 : 2;
  const int gx = (a.N + 64 * tn - 1) / (64 * tn), gy = (a.M + 64 * tm - 1) / (64 * tm);
  const int taps = a.gather && a.taps > 1 ? a.taps : 1;
  a.taps = taps;
  const long steps = (a.K + WG_BK - 1) / WG_BK;
  const long tiles = (long)gx * gy * taps;
  int ks = 1;
  if (ws && tiles < 1024 && steps >= 8) {
    ks = (int)((1024 + tiles - 1) / tiles);
    const long max_by_k = steps / 4 > 0 ? steps / 4 : 1;            // at least 4 slabs per slice
    if (ks > max_by_k) ks = (int)max_by_k;
    const size_t per = (size_t)a.M * a.N * taps;
    if ((size_t)ks * per > ws_floats) ks = (int)(ws_floats / per);
    if (ks < 1) ks = 1;
  }
  const long chunk = ((steps + ks - 1) / ks) * WG_BK;
  ks = (int)((a.K + chunk - 1) / chunk);
  if (taps > 1 && (!ws || (size_t)ks * a.M * a.N * taps > ws_floats)) return hipErrorInvalidValue;
  a.ksplit = ks; a.kchunk = chunk; a.ws = ws;
  const dim3 grid(gx, gy, ks * taps);
  if (tm == 2 && tn == 2) hipLaunchKernelGGL((wgrad_bf16_kernel<2, 2>), grid, dim3(256), 0, st, a);
  else if (tm == 2) hipLaunchKernelGGL((wgrad_bf16_kernel<2, 1>), grid, dim3(256), 0, st, a);
  else if (tn == 2) hipLaunchKernelGGL((wgrad_bf16_kernel<1, 2>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((wgrad_bf16_kernel<1, 1>), grid, dim3(256), 0, st, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (ks > 1 || taps > 1) {
    const size_t mn = (size_t)a.M * a.N * taps;
    const unsigned blocks = (unsigned)((mn + 255) / 256 < 4096 ? (mn + 255) / 256 : 4096);
    hipLaunchKernelGGL(sgemm_reduce_kernel, dim3(blocks), dim3(256), 0, st, ws, a.C, a.M, a.N, a.ldc, ks, a.accumulate, taps,
                       a.tapC, a.bias);
    e = hipGetLastError();
  }
  return e;
}

}  // namespace lrp
