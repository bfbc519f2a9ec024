// Dense fp32 products of the fine-tune step (train.py:573-581, models/model.py:1340-1374) on the fp32
// matrix cores: C (+)= op(A) . op(B), row-major operands with leading dimensions, any M / N / K.
//
//   TN  C = A^T B   weight gradients  dW = X^T dY, K = rows of the batch (B*T, B*L or B*H*W pixels) — both operands
//                   are read as they lie in memory (k-major rows); split over K with a deterministic second pass.
//                   With `gather` the rows of A are the pixels of an NHWC activation shifted by one 3x3 tap (zero
//                   outside the image): the weight gradient of a 'same' convolution, one launch per tap.
//   NN  C = A B     forward products X W          NT  C = A B^T   backward-data products dY W^T
//
// v_mfma_f32_32x32x2_f32: lane l feeds A[m = l & 31][k = l >> 5] and B[k = l >> 5][n = l & 31] — one scalar per lane
// from a k-major LDS tile, consecutive lanes on consecutive banks.  Tile 128 x 128 x 16, 4 waves of 64 x 64, register
// prefetch of the next k-slab while the current one is multiplied.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

namespace lrp {

struct SgemmArgs {
  const float* A; const float* B; float* C;
  int M, N; long K;
  long lda, ldb, ldc;
  int transA, transB;      // op(A)[m][k] = transA ? A[k*lda + m] : A[m*lda + k];  op(B)[k][n] = transB ? B[n*ldb + k] : B[k*ldb + n]
  int accumulate;          // C += result
  int ksplit; long kchunk; // blockIdx.z = slice of K; partials go to ws[slice][M][N] when ksplit > 1
  float* ws;
  int gather, gH, gW, dy, dx;   // transA only: row r = (n*gH + y)*gW + x reads row r + dy*gW + dx, zero outside the image
  int taps; long tapC;          // gather with taps = 9: blockIdx.z / ksplit = tap (dy, dx from it), C advances by tapC per tap
  // without gather, taps > 1 = a strided BATCH of products of one shape in one launch: product p reads A + p * batchA and
  // B + p * batchB and writes C + p * tapC (the four per-gate products of an LSTM step under keras' per-gate dropout masks)
  long batchA, batchB;
  int vecA, vecB;          // 16 B loads allowed (alignment checked on the host)
  const float* bias;       // optional: + bias[n] on the final result (not with accumulate)
  // second product of the same shape in the same launch (consumer-side reduction only, sgemm(..., ks_out)): the upper half
  // of blockIdx.z multiplies A2 . B2 into ws2 (two launch-bound products of a decoder step become one launch)
  const float* A2; const float* B2; float* ws2;
};

constexpr int SG_BK = 16;
constexpr int sg_ld(int bmn) { return bmn + 4; }       // LDS row stride (floats): keeps 16 B alignment, spreads the transposing writes

typedef float sg_f32x16 __attribute__((ext_vector_type(16)));

// stage one operand tile (16 k x BMN cols) into LDS [k][col]; kmajor: memory rows are k
template <bool KMAJOR, int BMN>
__device__ __forceinline__ void sg_fetch(const float* __restrict__ P, long ld, int rows_mn, long k_end, int mn0, long k0, int tid,
                                         bool vec, const SgemmArgs& a, bool is_a, float (&r)[BMN / 64][4]) {
#pragma unroll
  for (int q = 0; q < BMN / 64; ++q) {
    const int idx = tid + 256 * q;
    if constexpr (KMAJOR) {
      const int kr = idx / (BMN / 4), c4 = idx % (BMN / 4);
      const long k = k0 + kr;
      const int col = mn0 + 4 * c4;
      long src = k;
      bool ok = k < k_end;
      if (is_a && a.gather && ok) {
        const unsigned ku = (unsigned)k, rowi = ku / (unsigned)a.gW;      // (pixel counts stay far below 2^31)
        const int x = (int)(ku - rowi * (unsigned)a.gW), y = (int)(rowi % (unsigned)a.gH);
        ok = (unsigned)(y + a.dy) < (unsigned)a.gH && (unsigned)(x + a.dx) < (unsigned)a.gW;
        src = k + (long)a.dy * a.gW + a.dx;
      }
      if (ok && vec && col + 3 < rows_mn) {
        const float4 v = *reinterpret_cast<const float4*>(P + src * ld + col);
        r[q][0] = v.x; r[q][1] = v.y; r[q][2] = v.z; r[q][3] = v.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) r[q][j] = (ok && col + j < rows_mn) ? P[src * ld + col + j] : 0.f;
      }
    } else {
      const int row = idx >> 2, c4 = idx & 3;
      const int mn = mn0 + row;
      const long k = k0 + 4 * c4;
      const bool ok = mn < rows_mn;
      if (ok && vec && k + 3 < k_end) {
        const float4 v = *reinterpret_cast<const float4*>(P + (long)mn * ld + k);
        r[q][0] = v.x; r[q][1] = v.y; r[q][2] = v.z; r[q][3] = v.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) r[q][j] = (ok && k + j < k_end) ? P[(long)mn * ld + k + j] : 0.f;
      }
    }
  }
}
template <bool KMAJOR, int BMN>
__device__ __forceinline__ void sg_put(float* __restrict__ S, int tid, const float (&r)[BMN / 64][4]) {
  constexpr int LD = sg_ld(BMN);
#pragma unroll
  for (int q = 0; q < BMN / 64; ++q) {
    const int idx = tid + 256 * q;
    if constexpr (KMAJOR) {
      const int kr = idx / (BMN / 4), c4 = idx % (BMN / 4);
      *reinterpret_cast<float4*>(S + kr * LD + 4 * c4) = make_float4(r[q][0], r[q][1], r[q][2], r[q][3]);
    } else {
      const int row = idx >> 2, c4 = idx & 3;
#pragma unroll
      for (int j = 0; j < 4; ++j) S[(4 * c4 + j) * LD + row] = r[q][j];
    }
  }
}

// TM x TN 32x32 MFMA tiles per wave, 2 x 2 waves: block tile (64 TM) x (64 TN)
template <bool TA, bool TB, int TM, int TN>
__global__ __launch_bounds__(256) void sgemm_kernel(SgemmArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = 64 * TM, BN = 64 * TN, LDA = sg_ld(BM), LDB = sg_ld(BN);
  __shared__ float As[2][SG_BK * LDA];
  __shared__ float Bs[2][SG_BK * LDB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  int bz = (int)blockIdx.z;
  if (a.A2 && bz >= a.ksplit) {                        // the second product of a paired launch
    bz -= a.ksplit;
    a.A = a.A2; a.B = a.B2; a.ws = a.ws2; a.C = a.ws2;
  }
  const int tap = a.taps > 1 ? bz / a.ksplit : 0, slice = a.taps > 1 ? bz % a.ksplit : bz;
  if (a.taps > 1) {
    if (a.gather) { a.dy = tap / 3 - 1; a.dx = tap % 3 - 1; }
    else { a.A += (long)tap * a.batchA; a.B += (long)tap * a.batchB; }
  }
  const long kb = (long)slice * a.kchunk;
  const long ke = kb + a.kchunk < a.K ? kb + a.kchunk : a.K;
  sg_f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float ra[TM][4], rb[TN][4];
  const long nst = (ke - kb + SG_BK - 1) / SG_BK;
  if (nst > 0) {
    sg_fetch<TA, BM>(a.A, a.lda, a.M, ke, m0, kb, tid, a.vecA != 0, a, true, ra);
    sg_fetch<!TB, BN>(a.B, a.ldb, a.N, ke, n0, kb, tid, a.vecB != 0, a, false, rb);
    sg_put<TA, BM>(As[0], tid, ra);
    sg_put<!TB, BN>(Bs[0], tid, rb);
  }
  __syncthreads();
  for (long s = 0; s < nst; ++s) {
    const int cur = (int)(s & 1);
    const bool more = s + 1 < nst;
    if (more) {
      sg_fetch<TA, BM>(a.A, a.lda, a.M, ke, m0, kb + (s + 1) * SG_BK, tid, a.vecA != 0, a, true, ra);
      sg_fetch<!TB, BN>(a.B, a.ldb, a.N, ke, n0, kb + (s + 1) * SG_BK, tid, a.vecB != 0, a, false, rb);
    }
    const float* Ac = As[cur] + (lane >> 5) * LDA + wm * 32 * TM + (lane & 31);
    const float* Bc = Bs[cur] + (lane >> 5) * LDB + wn * 32 * TN + (lane & 31);
    // fragments of k-pair kk + 1 are read while the MFMAs of kk issue (two register sets)
    float av[2][TM], bv[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) av[0][i] = Ac[32 * i];
#pragma unroll
    for (int j = 0; j < TN; ++j) bv[0][j] = Bc[32 * j];
#pragma unroll
    for (int kk = 0; kk < SG_BK / 2; ++kk) {
      const int c = kk & 1, nx = c ^ 1;
      if (kk + 1 < SG_BK / 2) {
#pragma unroll
        for (int i = 0; i < TM; ++i) av[nx][i] = Ac[2 * (kk + 1) * LDA + 32 * i];
#pragma unroll
        for (int j = 0; j < TN; ++j) bv[nx][j] = Bc[2 * (kk + 1) * LDB + 32 * j];
      }
      __builtin_amdgcn_sched_barrier(0);            // keep the reads above the MFMAs (hipcc sinks them to their first use)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c][i], bv[c][j], acc[i][j], 0, 0, 0);
    }
    if (more) {
      sg_put<TA, BM>(As[cur ^ 1], tid, ra);
      sg_put<!TB, BN>(Bs[cur ^ 1], tid, rb);
    }
    __syncthreads();
  }
  const bool partial = a.ksplit > 1 || a.taps > 1;     // through the workspace; sgemm_reduce_kernel finishes
  float* C = partial ? a.ws + (size_t)bz * a.M * a.N : a.C;
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
          const float v = acc[i][j][r] + ((a.bias && !partial) ? a.bias[col] : 0.f);
          *p = accum ? *p + v : v;
        }
      }
    }
#endif
}

// C[tap][m][n] (+)= sum_s ws[tap][s][m][n], slices added in index order
__global__ __launch_bounds__(256) void sgemm_reduce_kernel(const float* __restrict__ ws, float* __restrict__ C, int M, int N,
                                                           long ldc, int ksplit, int accumulate, int taps, long tapC,
                                                           const float* __restrict__ bias) {
  const size_t mn = (size_t)M * N, all = mn * taps;
  for (size_t j = (size_t)blockIdx.x * 256 + threadIdx.x; j < all; j += (size_t)gridDim.x * 256) {
    const size_t tap = j / mn, i = j - tap * mn;
    const float* w = ws + tap * ksplit * mn + i;
    float s = 0.f;
    for (int k = 0; k < ksplit; ++k) s += w[(size_t)k * mn];
    if (bias) s += bias[i % N];
    float* p = C + tap * tapC + (long)(i / N) * ldc + (i % N);
    *p = accumulate ? *p + s : s;
  }
}

inline bool sg_aligned(const void* p, long ld) { return (((uintptr_t)p) & 15) == 0 && (ld & 3) == 0; }

template <bool TA, bool TB>
inline void sgemm_launch_tile(int tm, int tn, dim3 grid, hipStream_t st, const SgemmArgs& a) {
  if (tm == 4) hipLaunchKernelGGL((sgemm_kernel<TA, TB, 4, 2>), grid, dim3(256), 0, st, a);
  else if (tn == 4) hipLaunchKernelGGL((sgemm_kernel<TA, TB, 2, 4>), grid, dim3(256), 0, st, a);
  else if (tm == 1 && tn == 1) hipLaunchKernelGGL((sgemm_kernel<TA, TB, 1, 1>), grid, dim3(256), 0, st, a);
  else if (tm == 1) hipLaunchKernelGGL((sgemm_kernel<TA, TB, 1, 2>), grid, dim3(256), 0, st, a);
  else if (tn == 1) hipLaunchKernelGGL((sgemm_kernel<TA, TB, 2, 1>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((sgemm_kernel<TA, TB, 2, 2>), grid, dim3(256), 0, st, a);
}

// ws / ws_floats: workspace for the K split (may be null: no split).  Tile 64 or 128 per side by the extent; the K
// split fills the chip when the tiles alone do not (weight gradients: few tiles, K = millions of pixels; the
// per-step products of the scan: 32 rows, a handful of tiles).
// ks_out != nullptr ("consumer-side reduction"): no reduce launch — the product is left as *ks_out slabs of M x N floats
// (row stride N) at ws, bias NOT added; the consumer sums the slabs in index order and adds the bias, which is exactly what
// sgemm_reduce_kernel would have done, one launch earlier.  (Needs ws; taps must be 1.)
inline hipError_t sgemm(SgemmArgs a, float* ws, size_t ws_floats, hipStream_t st, int* ks_out = nullptr) {
  if (a.M < 1 || a.N < 1 || a.K < 1) return hipErrorInvalidValue;
  if (ks_out) {
    if (!ws || (size_t)a.M * a.N > ws_floats || a.taps > 1 || a.accumulate) return hipErrorInvalidValue;
    a.C = ws; a.ldc = a.N; a.bias = nullptr;           // (a single slice writes straight to slab 0)
    if (a.A2 && (!a.B2 || !a.ws2 || !sg_aligned(a.A2, a.lda) != !sg_aligned(a.A, a.lda) || !sg_aligned(a.B2, a.ldb) != !sg_aligned(a.B, a.ldb)))
      return hipErrorInvalidValue;                     // (the pair shares one set of launch parameters)
  } else if (a.A2) {
    return hipErrorInvalidValue;
  }
  const bool batch = !a.gather && a.taps > 1;
  a.vecA = sg_aligned(a.A, a.lda) && (!batch || !(a.batchA & 3)) ? 1 : 0;
  a.vecB = sg_aligned(a.B, a.ldb) && (!batch || !(a.batchB & 3)) ? 1 : 0;
  int tm = a.M <= 64 ? 1 : 2, tn = a.N <= 64 ? 1 : 2;
  if (a.K >= 4096) {                                       // 256 x 128 tiles: 1.5x the FLOPs per byte staged
    if (a.M >= 256 && a.N >= 128) tm = 4;
    else if (a.N >= 256 && a.M >= 128) tn = 4;
  }
  const int gx = (a.N + 64 * tn - 1) / (64 * tn), gy = (a.M + 64 * tm - 1) / (64 * tm);
  int ks = 1;
  const int taps = a.taps > 1 ? a.taps : 1;             // nine gathered taps, or a strided batch
  a.taps = taps;
  const long steps = (a.K + SG_BK - 1) / SG_BK;
  const long tiles = (long)gx * gy * taps;
  if (ws && tiles < 384 && steps >= 16) {
    // ~3 workgroups per CU in flight; the 64 x 64 tile (17 KB of LDS, 4 MFMAs per wave and k-pair) is latency-bound per
    // k-slab, so its products — the image layer's weight gradient: M = N = 64, K = 1.6 M pixels — get ~8 per CU instead
    // [MI355X: 1.75 -> see profiles/r02_config5.txt]
    const long want = (tm == 1 && tn == 1) ? 2048 : 768;
    ks = (int)((want + tiles - 1) / tiles);
    const long max_by_k = steps / 8 > 0 ? steps / 8 : 1;          // at least 8 slabs per slice
    if (ks > max_by_k) ks = (int)max_by_k;
    const size_t per = (size_t)a.M * a.N * taps;
    if ((size_t)ks * per > ws_floats) ks = (int)(ws_floats / per);
    if (ks < 1) ks = 1;
  }
  long chunk = ((steps + ks - 1) / ks) * SG_BK;
  ks = (int)((a.K + chunk - 1) / chunk);
  // (the tap-folded launch always goes through the partials, also with a single K slice)
  if (taps > 1 && (!ws || (size_t)ks * a.M * a.N * taps > ws_floats)) return hipErrorInvalidValue;
  a.ksplit = ks; a.kchunk = chunk; a.ws = ws;
  dim3 grid(gx, gy, ks * taps * (a.A2 ? 2 : 1));
  if (a.transA && !a.transB) sgemm_launch_tile<true, false>(tm, tn, grid, st, a);
  else if (!a.transA && !a.transB) sgemm_launch_tile<false, false>(tm, tn, grid, st, a);
  else if (!a.transA && a.transB) sgemm_launch_tile<false, true>(tm, tn, grid, st, a);
  else return hipErrorInvalidValue;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (ks_out) { *ks_out = ks; return e; }
  if (ks > 1 || taps > 1) {
    const size_t mn = (size_t)a.M * a.N * taps;
    const unsigned blocks = (unsigned)((mn + 255) / 256 < 4096 ? (mn + 255) / 256 : 4096);
    hipLaunchKernelGGL(sgemm_reduce_kernel, dim3(blocks), dim3(256), 0, st, ws, a.C, a.M, a.N, a.ldc, ks, a.accumulate, taps,
                       a.tapC, a.bias);
    e = hipGetLastError();
  }
  return e;
}

// column sums of a (rows x N) matrix: out[n] (+)= sum_r X[r][n]; two deterministic passes through `part` (chunks x N)
__global__ __launch_bounds__(256) void colsum_part_kernel(const float* __restrict__ X, long ld, long rows, int N, long rows_per,
                                                          float* __restrict__ part) {
  __shared__ float red[256];
  const long r0 = (long)blockIdx.x * rows_per;
  const long r1 = r0 + rows_per < rows ? r0 + rows_per : rows;
  // columns in passes of up to 256; with fewer columns the spare threads take interleaved rows (4 loads in flight each)
  const int cw = N < 256 ? N : 256, lanes = 256 / cw, tid = threadIdx.x;
  for (int n0 = 0; n0 < N; n0 += cw) {
    const int c = tid % cw, rl = tid / cw, n = n0 + c;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (rl < lanes && n < N) {
      long r = r0 + rl;
      for (; r + 3L * lanes < r1; r += 4L * lanes) {
        s0 += X[r * ld + n]; s1 += X[(r + lanes) * ld + n]; s2 += X[(r + 2L * lanes) * ld + n]; s3 += X[(r + 3L * lanes) * ld + n];
      }
      for (; r < r1; r += lanes) s0 += X[r * ld + n];
    }
    float s = (s0 + s1) + (s2 + s3);
    red[tid] = s;
    __syncthreads();
    if (rl == 0 && n < N) {
      for (int j = 1; j < lanes; ++j) s += red[j * cw + c];
      part[(size_t)blockIdx.x * N + n] = s;
    }
    __syncthreads();
  }
}
// one wave per column: lanes take interleaved chunks, fixed-order tree over the lanes
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, int chunks, int N, float* __restrict__ out,
                                                           int accumulate) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (n >= N) return;
  float s = 0.f;
  for (int c = lane; c < chunks; c += 64) s += part[(size_t)c * N + n];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) out[n] = accumulate ? out[n] + s : s;
}
inline hipError_t colsum(const float* X, long ld, long rows, int N, float* out, int accumulate, float* ws, size_t ws_floats,
                         hipStream_t st) {
  long chunks = (rows + 255) / 256;
  if (chunks > 2048) chunks = 2048;
  if ((size_t)chunks * N > ws_floats) chunks = (long)(ws_floats / N);
  if (chunks < 1) return hipErrorInvalidValue;
  const long per = (rows + chunks - 1) / chunks;
  chunks = (rows + per - 1) / per;
  hipLaunchKernelGGL(colsum_part_kernel, dim3((unsigned)chunks), dim3(256), 0, st, X, ld, rows, N, per, ws);
  hipLaunchKernelGGL(colsum_final_kernel, dim3((N + 3) / 4), dim3(256), 0, st, ws, (int)chunks, N, out, accumulate);
  return hipGetLastError();
}

}  // namespace lrp
