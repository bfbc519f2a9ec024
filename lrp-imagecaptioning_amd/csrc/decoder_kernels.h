// decoder_kernels.h — HIP kernels of the decoder half (adaptive attention):
//   forward replay   _forward_beam_search            E:370-436   (E: = models/explainers.py)
//   per-token LRP    _explain_lstm_single_word_sequence  E:537-666, rule E:156-165
// Precision mirrors the reference: float32 LSTM / attention chain, float64 from
// `context` on and for every LRP accumulator, float32 stores into r_V / R_feat.
// These kernels are HBM/L2- or latency-bound (GEMV-like); the design rules that
// matter are coalesced weight streams (weights pre-transposed so that lanes run
// along the contiguous axis) and keeping the per-token scan inside ONE launch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lrp {

// dst[c * ldd + r] = src[r * lds + c]   (32 x 32 tiles through LDS; block (32, 8))
__global__ void dec_transpose_kernel(const float* __restrict__ src, int lds, int rows, int cols, float* __restrict__ dst, int ldd) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int j = threadIdx.y; j < 32; j += 8) {
    const int r = r0 + j, c = c0 + threadIdx.x;
    tile[j][threadIdx.x] = (r < rows && c < cols) ? src[(size_t)r * lds + c] : 0.f;
  }
  __syncthreads();
  for (int j = threadIdx.y; j < 32; j += 8) {
    const int c = c0 + j, r = r0 + threadIdx.x;
    if (c < cols && r < rows) dst[(size_t)c * ldd + r] = tile[threadIdx.x][j];
  }
}


constexpr double LRP_EPS = 1e-7;     // K.epsilon() bound at E:157

// z + sign(z)*eps, sign(0) = +1   (E:141-144)
__device__ __forceinline__ double stab(double z) { return z + (z < 0.0 ? -LRP_EPS : LRP_EPS); }
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// ------------------------------------------------------------------------------------------
// Skinny GEMM: Y[r][n] = act( sum_k X[r][k] * W[k][n] + bias[n] ),  R = a handful of rows
// (images or image*step), W in the Keras (in,out) layout -> lanes run along n (coalesced).
// Block = 64 columns x 32 rows, 4 waves split K; partial sums meet in LDS.
// ------------------------------------------------------------------------------------------
// Split-K: blockIdx.z owns K-range [z*kchunk, (z+1)*kchunk) and writes its partial sums to slab z
// (Y + z*slab); the consumer kernel adds the slabs in a fixed order (bit-reproducible, no atomics).
// These GEMMs are latency-bound chains of dependent loads, so more, shorter blocks is the lever.
template <typename TX, typename TA, typename TY>
__device__ __forceinline__ void skinny_gemm_body(const TX* __restrict__ X, int ldx, const float* __restrict__ W, int ldw,
                                                 const float* __restrict__ bias, TY* __restrict__ Y, int ldy, int R, int K, int N,
                                                 int relu, int kchunk, size_t slab, int row_block) {
  __shared__ TA xs[64][32];
  __shared__ TA red[3][32][64];
  const int tid = threadIdx.x, col = tid & 63, kg = tid >> 6;
  const int n = blockIdx.x * 64 + col, r0 = row_block * 32;
  const int kbeg = blockIdx.z * kchunk;
  const int Kend = (kbeg + kchunk < K) ? kbeg + kchunk : K;
  Y += (size_t)blockIdx.z * slab;
  if (blockIdx.z) bias = nullptr;
  TA acc[32];
#pragma unroll
  for (int r = 0; r < 32; ++r) acc[r] = (TA)0;
  for (int k0 = kbeg; k0 < Kend; k0 += 64) {
    for (int e = tid; e < 2048; e += 256) {
      const int k = e & 63, r = e >> 6;
      TA v = (TA)0;
      if (r0 + r < R && k0 + k < Kend) v = (TA)X[(size_t)(r0 + r) * ldx + k0 + k];
      xs[k][r] = v;
    }
    __syncthreads();
    if (n < N) {
      if (k0 + 64 <= Kend) {
        // whole slab: the 16 weight loads of this wave's k rows are requested together (guarding each load made hipcc
        // branch around it and wait for it alone: a chain of 16 dependent L2 round trips per slab)
        float wv[16];
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) wv[kk] = W[(size_t)(k0 + kg * 16 + kk) * ldw + n];
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
          const TA w = (TA)wv[kk];
#pragma unroll
          for (int r = 0; r < 32; ++r) acc[r] += xs[kg * 16 + kk][r] * w;
        }
      } else {
#pragma unroll 4
        for (int kk = 0; kk < 16; ++kk) {
          const int k = kg * 16 + kk, gk = k0 + k;
          if (gk < Kend) {
            const TA w = (TA)W[(size_t)gk * ldw + n];
#pragma unroll
            for (int r = 0; r < 32; ++r) acc[r] += xs[k][r] * w;
          }
        }
      }
    }
    __syncthreads();
  }
  if (kg > 0) {
#pragma unroll
    for (int r = 0; r < 32; ++r) red[kg - 1][r][col] = acc[r];
  }
  __syncthreads();
  if (kg == 0 && n < N) {
    const TA b = bias ? (TA)bias[n] : (TA)0;
#pragma unroll
    for (int r = 0; r < 32; ++r) {
      if (r0 + r < R) {
        TA v = acc[r] + red[0][r][col] + red[1][r][col] + red[2][r][col] + b;
        if (relu) v = v > (TA)0 ? v : (TA)0;
        Y[(size_t)(r0 + r) * ldy + n] = (TY)v;
      }
    }
  }
}
template <typename TX, typename TA, typename TY>
__global__ __launch_bounds__(256) void skinny_gemm_kernel(const TX* __restrict__ X, int ldx, const float* __restrict__ W,
                                                          int ldw, const float* __restrict__ bias, TY* __restrict__ Y,
                                                          int ldy, int R, int K, int N, int relu, int kchunk, size_t slab) {
  skinny_gemm_body<TX, TA, TY>(X, ldx, W, ldw, bias, Y, ldy, R, K, N, relu, kchunk, slab, blockIdx.y);
}
// TWO products of one shape in one launch (R <= 32 rows: blockIdx.y picks the product) — the grid-TD step's h1 . W_ha and
// s . W_s, which were two dependent-looking launches of ~25 us each on a latency-bound chain
struct SkinnyPair { const void* X[2]; const float* W[2]; void* Y[2]; };
template <typename TX, typename TA, typename TY>
__global__ __launch_bounds__(256) void skinny_gemm_pair_kernel(SkinnyPair p, int ldx, int ldw, int ldy, int R, int K, int N,
                                                               int kchunk, size_t slab) {
  const int w = blockIdx.y;
  skinny_gemm_body<TX, TA, TY>(static_cast<const TX*>(p.X[w]), ldx, p.W[w], ldw, nullptr, static_cast<TY*>(p.Y[w]), ldy, R, K, N, 0,
                               kchunk, slab, 0);
}

// Caption generation (beam search) re-parents hypotheses between steps: row r continues the hypothesis that lived in
// row parent[r].  Two phases (all reads, then all writes) so that a row may be both a source and a destination.
template <typename T>
__global__ __launch_bounds__(256) void gen_gather_kernel(const T* __restrict__ arr, T* __restrict__ tmp,
                                                         const int* __restrict__ parent, int step, int S, int H) {
  const int r = blockIdx.x;
  for (int j = threadIdx.x; j < H; j += 256) tmp[(size_t)r * H + j] = arr[((size_t)parent[r] * S + step) * H + j];
}
template <typename T>
__global__ __launch_bounds__(256) void gen_scatter_kernel(T* __restrict__ arr, const T* __restrict__ tmp, int step, int S, int H) {
  const int r = blockIdx.x;
  for (int j = threadIdx.x; j < H; j += 256) arr[((size_t)r * S + step) * H + j] = tmp[(size_t)r * H + j];
}
__global__ void gen_set_word_kernel(int* __restrict__ cap, const int* __restrict__ word, int step, int Tm, int B) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r < B) cap[(size_t)r * Tm + step - 1] = word[r];
}

// ipre = 1 / stab(if_pre): the denominator of the image_features rule (E:654-659), once per image
__global__ __launch_bounds__(256) void dec_ipre_kernel(const float* __restrict__ if_pre, double* __restrict__ ipre, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    ipre[i] = 1.0 / stab((double)if_pre[i]);
}

// avg[b][d] = mean_l F[b][l][d]   (np.mean(axis=0) in float32: sequential row adds, then / L; E:382)
// grid (B, ceil(D / 64)), 64 threads: one channel per thread, rows added in order; the loads of 14 rows are issued together
// (one image, one 256-thread block walking 196 dependent rows: 55 us [MI355X])
__global__ __launch_bounds__(64) void mean_rows_kernel(const float* __restrict__ F, float* __restrict__ avg, int L, int D) {
  const int b = blockIdx.x, d = blockIdx.y * 64 + threadIdx.x;
  if (d >= D) return;
  const float* f = F + (size_t)b * L * D + d;
  float s = 0.f;
  constexpr int U = 14;
  int l0 = 0;
  for (; l0 + U <= L; l0 += U) {                        // (whole batches unconditionally: a guard per load makes hipcc wait per load)
    float v[U];
#pragma unroll
    for (int q = 0; q < U; ++q) v[q] = f[(size_t)(l0 + q) * D];
#pragma unroll
    for (int q = 0; q < U; ++q) s += v[q];
  }
  for (; l0 < L; ++l0) s += f[(size_t)l0 * D];
  avg[(size_t)b * D + d] = s / (float)L;
}

// Step prologue: xh[b] = [ embedding(tok) | relu(glob_pre_b) | h_{i} ]  and xt[b][i] = first 2E   (E:386, E:402-409)
__global__ __launch_bounds__(256) void dec_prep_x_kernel(const float* __restrict__ emb, const float* __restrict__ glob_pre,
                                                         const float* __restrict__ ht, const int* __restrict__ cap,
                                                         float* __restrict__ xh, float* __restrict__ xt, int step, int Tm,
                                                         int E, int H, int V, int sos) {
  const int b = blockIdx.x, S = Tm + 1;
  int tok = (step == 0 ? sos : cap[b * Tm + step - 1]) - 1;
  tok = tok < 0 ? 0 : (tok >= V ? V - 1 : tok);
  const int Nd = 2 * E + H;
  for (int e = threadIdx.x; e < Nd; e += 256) {
    float v;
    if (e < E) v = emb[(size_t)tok * E + e];
    else if (e < 2 * E) v = fmaxf(glob_pre[(size_t)b * E + e - E], 0.f);
    else v = ht[((size_t)b * S + step) * H + e - 2 * E];
    xh[(size_t)b * Nd + e] = v;
    if (e < 2 * E) xt[((size_t)b * Tm + step) * 2 * E + e] = v;
  }
}

// LSTM pointwise (E:129-138) + visual sentinel s = tanh(c) * sigmoid(x.Wx + h_prev.Wh) (E:415).
// z[b] = [ i | f | g | o | sentinel-gate ] pre-activations (5H).  Writes state row step+1.
__global__ __launch_bounds__(256) void dec_pointwise_kernel(const float* __restrict__ z, int ks, size_t slab,
                                                            const float* __restrict__ bias, float* __restrict__ ht,
                                                            float* __restrict__ ct, float* __restrict__ gt,
                                                            float* __restrict__ it, float* __restrict__ ft,
                                                            float* __restrict__ st, float* __restrict__ ot, int step,
                                                            int Tm, int H) {
  // grid (B, ceil(H / blockDim.x)): one hidden unit per thread; the slabs of the five pre-activations are added in index
  // order, four slabs (20 loads) requested at a time
  const int b = blockIdx.x, S = Tm + 1;
  const float* zb = z + (size_t)b * 5 * H;
  const size_t prev = ((size_t)b * S + step) * H, cur = prev + H;
  const int j = blockIdx.y * blockDim.x + threadIdx.x;
  if (j < H) {
    float zz[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    int q = 0;
    for (; q + 4 <= ks; q += 4) {
      float t[4][5];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int g = 0; g < 5; ++g) t[u][g] = zb[(size_t)(q + u) * slab + g * H + j];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int g = 0; g < 5; ++g) zz[g] += t[u][g];
    }
    for (; q < ks; ++q)
#pragma unroll
      for (int g = 0; g < 5; ++g) zz[g] += zb[(size_t)q * slab + g * H + j];          // split-K slabs, fixed order
    if (bias) {                                          // (slabs straight from sgemm: the bias joins after the slices, as in sgemm_reduce_kernel)
#pragma unroll
      for (int g = 0; g < 5; ++g) zz[g] += bias[g * H + j];
    }
    const float i_ = sigmoidf_(zz[0]), f_ = sigmoidf_(zz[1]), g_ = zz[2], o_ = sigmoidf_(zz[3]);
    const float c = f_ * ct[prev + j] + i_ * tanhf(g_);
    const float tc = tanhf(c);
    ht[cur + j] = o_ * tc;
    ct[cur + j] = c;
    gt[cur + j] = g_;
    it[cur + j] = i_;
    ft[cur + j] = f_;
    ot[cur + j] = o_;                                  // (read by the gradient baselines only, E:673-688)
    st[cur + j] = tc * sigmoidf_(zz[4]);
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Attention softmax + sentinel mix for one step (E:412-421), two launches so that more than B
// workgroups are in flight:
//   scores:  pre[b][l] = tanh(h.Wg + V.Wv)[l] . v  for l < L,  pre[b][L] = tanh(s.Ws + h.Wg) . v
//            grid (B, ceil((L+1)/ROWS)), one wave per row, lanes along H      float32
//   finish:  alpha = softmax_L(pre), beta = last entry of softmax over [pre ; sentinel],
//            ctx = sum_l alpha_l relu(if_pre_l) (float64), c_hat = beta*s + (1-beta)*ctx, u = h + c_hat
// hproj / sproj arrive as split-K slabs (summed here in fixed order).
constexpr int ATT_ROWS = 16;
__global__ __launch_bounds__(256) void dec_att_scores_kernel(const float* __restrict__ hproj, const float* __restrict__ sproj,
                                                             int ks, size_t slab, const float* __restrict__ stat,
                                                             const float* __restrict__ vvec, float* __restrict__ pre,
                                                             int L, int H) {
  extern __shared__ float fsm[];
  float* hp = fsm;
  float* sp = fsm + H;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l0 = blockIdx.y * ATT_ROWS;
  const bool need_s = l0 + ATT_ROWS > L;                 // only the block that owns row L needs s.Ws
  for (int j = tid; j < H; j += 256) {
    float h = 0.f, sv = 0.f;
    for (int q = 0; q < ks; ++q) {
      h += hproj[(size_t)q * slab + (size_t)b * H + j];
      if (need_s) sv += sproj[(size_t)q * slab + (size_t)b * H + j];
    }
    hp[j] = h;
    sp[j] = sv;
  }
  __syncthreads();
  for (int l = l0 + wave; l < l0 + ATT_ROWS && l <= L; l += 4) {
    float p = 0.f;
    if (l < L) {
      const float* srow = stat + ((size_t)b * L + l) * H;
      for (int j = lane; j < H; j += 64) p += tanhf(hp[j] + srow[j]) * vvec[j];
    } else {
      for (int j = lane; j < H; j += 64) p += tanhf(sp[j] + hp[j]) * vvec[j];
    }
    p = wave_sum(p);
    if (lane == 0) pre[(size_t)b * (L + 1) + l] = p;
  }
}

// dynamic LDS: float pre[L+1]
__global__ __launch_bounds__(256) void dec_att_finish_kernel(const float* __restrict__ pre_g, const float* __restrict__ if_pre,
                                                             const float* __restrict__ ht, const float* __restrict__ st,
                                                             float* __restrict__ att, float* __restrict__ beta,
                                                             double* __restrict__ ctx, double* __restrict__ chat,
                                                             double* __restrict__ u, int step, int Tm, int L, int H) {
  // grid (B, ceil(H / 64)), 64 threads: every block redoes the (cheap) soft-max over its image's L + 1 scores and owns 64
  // channels of the context sum; block y = 0 also stores attention / beta.  [MI355X, one image: one 256-thread block per image
  // walked L rows for two channels per thread: 20 us per step]
  extern __shared__ float fsm[];
  float* pre = fsm;
  const int b = blockIdx.x, S = Tm + 1, tid = threadIdx.x, lane = tid & 63;
  const bool first = blockIdx.y == 0;
  for (int l = tid; l <= L; l += 64) pre[l] = pre_g[(size_t)b * (L + 1) + l];
  __syncthreads();
  const size_t row = (size_t)b * S + step + 1;
  {
    float mx = -INFINITY;
    for (int l = lane; l < L; l += 64) mx = fmaxf(mx, pre[l]);
    mx = wave_max(mx);
    float sm = 0.f;
    for (int l = lane; l < L; l += 64) sm += expf(pre[l] - mx);
    sm = wave_sum(sm);
    const float zt = pre[L];
    const float mx2 = fmaxf(mx, zt);
    float sm2 = 0.f;
    for (int l = lane; l < L; l += 64) sm2 += expf(pre[l] - mx2);
    sm2 = wave_sum(sm2);
    const float ez = expf(zt - mx2);
    const float bt = ez / (sm2 + ez);
    for (int l = lane; l < L; l += 64) {
      const float al = expf(pre[l] - mx) / sm;
      pre[l] = al;
      if (first) att[row * L + l] = al;
    }
    if (lane == 0) { pre[L] = bt; if (first) beta[row] = bt; }
  }
  __syncthreads();
  const float bt = pre[L];
  const int j = blockIdx.y * 64 + tid;
  if (j < H) {
    double c = 0.0;
    const float* ip = if_pre + (size_t)b * L * H + j;
    constexpr int U = 14;
    int l0 = 0;
    for (; l0 + U <= L; l0 += U) {
      float v[U];
#pragma unroll
      for (int q = 0; q < U; ++q) v[q] = ip[(size_t)(l0 + q) * H];
#pragma unroll
      for (int q = 0; q < U; ++q) c += (double)pre[l0 + q] * (double)fmaxf(v[q], 0.f);
    }
    for (; l0 < L; ++l0) c += (double)pre[l0] * (double)fmaxf(ip[(size_t)l0 * H], 0.f);
    const float s = st[row * H + j];
    const double ch = (double)(bt * s) + (double)(1.f - bt) * c;
    ctx[row * H + j] = c;
    chat[row * H + j] = ch;
    u[((size_t)b * Tm + step) * H + j] = (double)ht[row * H + j] + ch;
  }
}

// ------------------------------------------------------------------------------------------
// Per-token LRP through the decoder, one workgroup per (image, t) pair.  Closed form of
// E:537-666: identity-weight rule calls collapse to element-wise
// shares  part / stab(whole) * R ; the gate-g rule is a (2E+H) x H GEMV per scan step.
// WgT[j][d] = [Wi;Wh][d][2H+j] (transposed gate-g block), WglobT[e][d] = W_glob[d][e].
// Outputs per token: rctx (H) , ravg (D)  -> consumed by dec_tail_kernel; r_words.
// dynamic LDS (doubles): rc[H] rh[H] q[max(H,E)] rglob[E] red[4]
// ------------------------------------------------------------------------------------------
constexpr int SCAN_MAXR = 8;         // (2E+H) <= 8*256

struct ExplainArgs {
  const int* img_idx; const int* tpos;            // [n]
  const int* cap;                                 // [B][Tm]
  const float *ht, *ct, *gt, *it, *ft, *st, *beta, *att, *xt;
  const double *ctx, *chat, *preds;
  const float* Wout;                              // [H][V]
  const float* WgT;                               // [H][2E+H]
  const float* WglobT;                            // [E][D]
  const float *avg, *glob_pre;
  double *rctx, *ravg;                            // rho = r_ctx/stab(ctx) [n][H], r_avg [n][D]
  float* att_out;                                 // [n][L] or null
  double* rwords_out;                             // [n][Tm] or null
  int Tm, L, D, H, E, V, single_step;
};

__device__ __forceinline__ double block_sum_d(double v, double* red) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  v = wave_sum_d(v);
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  const double r = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(256) void dec_explain_adaptive_kernel(ExplainArgs a) {
  extern __shared__ double dsm[];
  const int H = a.H, E = a.E, D = a.D, Tm = a.Tm, S = Tm + 1;
  double* rc = dsm;
  double* rh = rc + H;
  double* q = rh + H;
  double* rglob = q + (H > E ? H : E);
  double* red = rglob + E;
  const int n = blockIdx.x, tid = threadIdx.x;
  const int b = a.img_idx[n], t = a.tpos[n];
  const int Nd = 2 * E + H;
  const size_t rowt = (size_t)b * S + t;

  // ---- output layer, h / c_hat split, context / sentinel split   (E:552-602)
  const int k = a.cap[b * Tm + t - 1] - 1;
  const double zk = a.preds[((size_t)b * Tm + (t - 1)) * a.V + k];
  const double bt32_1m = (double)(1.f - a.beta[rowt]);
  const float btf = a.beta[rowt];
  for (int j = tid; j < H; j += 256) {
    const double h = (double)a.ht[rowt * H + j], ch = a.chat[rowt * H + j];
    const double u = h + ch;
    const double r_u = ((double)a.Wout[(size_t)j * a.V + k] * u) / stab(zk) * zk;
    const double su = stab(u);
    const double r_h = h / su * r_u;
    const double r_ch = ch / su * r_u;
    const double sch = stab(ch);
    const double r_ctx = (bt32_1m * a.ctx[rowt * H + j]) / sch * r_ch;
    const double r_s = (double)(btf * a.st[rowt * H + j]) / sch * r_ch;
    a.rctx[(size_t)n * H + j] = r_ctx / stab(a.ctx[rowt * H + j]);      // rho_j, consumed by dec_tail_kernel
    rc[j] = r_s;                                  // r_ct[t] = r_st            (E:602)
    rh[j] = r_h;
  }
  for (int e = tid; e < E; e += 256) rglob[e] = 0.0;
  if (a.att_out)
    for (int l = tid; l < a.L; l += 256) a.att_out[(size_t)n * a.L + l] = a.att[rowt * a.L + l];
  if (a.rwords_out)
    for (int i = tid; i < Tm; i += 256) a.rwords_out[(size_t)n * Tm + i] = 0.0;
  __syncthreads();

  // ---- reverse scan over the LSTM steps   (E:604-632)
  const int i_stop = a.single_step ? t - 1 : 0;
  for (int i = t - 1; i >= i_stop; --i) {
    const size_t r1 = ((size_t)b * S + i + 1) * H, r0 = ((size_t)b * S + i) * H;
    for (int j = tid; j < H; j += 256) {
      const double rcj = rc[j] + rh[j];                                   // r_ct[i+1] += r_ht[i+1]
      const double sc = stab((double)a.ct[r1 + j]);
      const float pg = a.it[r1 + j] * tanhf(a.gt[r1 + j]);                // float32 product, as in numpy
      const float pc = a.ft[r1 + j] * a.ct[r0 + j];
      const double r_g = (double)pg / sc * rcj;
      rc[j] = (double)pc / sc * rcj;                                      // r_ct[i]
      q[j] = r_g / stab((double)a.gt[r1 + j]);
    }
    __syncthreads();
    // GEMV acc[d] = sum_j WgT[j][d] * q[j].  Column indices are clamped (not branched) so the
    // loads of 4 consecutive j are independent and all in flight together; lanes run along d
    // (coalesced 256 B per wave-load).
    double acc[SCAN_MAXR];
    int dcl[SCAN_MAXR];
#pragma unroll
    for (int r = 0; r < SCAN_MAXR; ++r) { acc[r] = 0.0; const int d = tid + 256 * r; dcl[r] = d < Nd ? d : Nd - 1; }
    const int nr = (Nd + 255) >> 8;                       // live accumulators (uniform)
    int j = 0;
    for (; j + 4 <= H; j += 4) {
      float wv[4][SCAN_MAXR];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int r = 0; r < SCAN_MAXR; ++r)
          if (r < nr) wv[u][r] = a.WgT[(size_t)(j + u) * Nd + dcl[r]];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const double qj = q[j + u];
#pragma unroll
        for (int r = 0; r < SCAN_MAXR; ++r)
          if (r < nr) acc[r] += (double)wv[u][r] * qj;
      }
    }
    for (; j < H; ++j) {
      const double qj = q[j];
#pragma unroll
      for (int r = 0; r < SCAN_MAXR; ++r)
        if (r < nr) acc[r] += (double)a.WgT[(size_t)j * Nd + dcl[r]] * qj;
    }
    double wsum = 0.0;
#pragma unroll
    for (int r = 0; r < SCAN_MAXR; ++r) {
      const int d = tid + 256 * r;
      if (d < Nd) {
        const float x = d < 2 * E ? a.xt[((size_t)b * Tm + i) * 2 * E + d] : a.ht[r0 + d - 2 * E];
        const double rx = (double)x * acc[r];
        if (d < E) wsum += rx;                                            // r_wording_embedding[i]
        else if (d < 2 * E) rglob[d - E] += rx;                           // r_global_img_feature +=
        else rh[d - 2 * E] = rx;                                          // r_ht[i] =   ('=' E:627)
      }
    }
    const double ws = block_sum_d(wsum, red);                             // includes the barriers the next step needs
    if (tid == 0 && a.rwords_out) a.rwords_out[(size_t)n * Tm + i] = ws;
  }
  __syncthreads();

  // ---- global-feature rule  (E:634-639):  r_avg = avg * ( W_glob . (r_glob / stab(glob_pre)) )
  for (int e = tid; e < E; e += 256) q[e] = rglob[e] / stab((double)a.glob_pre[(size_t)b * E + e]);
  __syncthreads();
  for (int d = tid; d < D; d += 256) {
    double s = 0.0;
    for (int e = 0; e < E; ++e) s += (double)a.WglobT[(size_t)e * D + d] * q[e];
    a.ravg[(size_t)n * D + d] = (double)a.avg[(size_t)b * D + d] * s;
  }

  // ---- r_words post-processing (adaptive): [0] = 0, / max|.|, drop first   (E:660-665)
  if (a.rwords_out && !a.single_step) {
    __syncthreads();
    if (tid == 0) {
      double* rw = a.rwords_out + (size_t)n * Tm;
      rw[0] = 0.0;
      double m = 0.0;
      for (int i = 0; i < t; ++i) m = fmax(m, fabs(rw[i]));
      for (int i = 0; i + 1 < t; ++i) rw[i] = m != 0.0 ? rw[i + 1] / m : rw[i + 1];
      rw[t - 1] = 0.0;
    }
  }
}

// ------------------------------------------------------------------------------------------
// Tail of the per-token LRP (E:641-659): for every location l and feature d
//   r_V[l][j]   = float32( relu(if_pre[l][j]) * alpha_l / stab(ctx[j]) * r_ctx[j] )          (E:648-653)
//   R_feat[l][d]= float32( F[l][d]/L / stab(avg[d]) * r_avg[d] )                             (E:642-647)
//               + F[l][d] * sum_j W_if[d][j] * r_V[l][j] / stab(if_pre[l][j])                (E:654-659)
// = an (L x H) . (H x D) float64 GEMM per token whose A operand is generated on the fly.
// grid (n, ceil(L/64), ceil(D/64)); 256 threads, 4x4 outputs each; WifT[j][d] = W_if[d][j].
// ------------------------------------------------------------------------------------------
struct TailArgs {
  const int* img_idx; const int* tpos;
  const float* F;          // [B][L][D]
  const float* vfeat;      // [B][L][H]   relu(if_pre)
  const double* ipre;      // [B][L][H]   1 / stab(if_pre)   (per image, so the tail has no divides)
  const float* att;        // [B][S][L]
  const float* avg;        // [B][D]
  const float* WifT;       // [H][D]
  const double *rctx, *ravg;   // rctx holds rho = r_ctx / stab(ctx)
  float* R_feat;           // [n][L][D]
  int Tm, L, D, H;
};

__global__ __launch_bounds__(256) void dec_tail_kernel(TailArgs a) {
  __shared__ double As[16][65];
  __shared__ double Bs[16][65];
  __shared__ double rho[16];     // r_ctx / stab(ctx) is applied per j: keep both factors exact instead
  const int n = blockIdx.x, l0 = blockIdx.y * 64, d0 = blockIdx.z * 64;
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int b = a.img_idx[n], t = a.tpos[n], S = a.Tm + 1;
  const int L = a.L, D = a.D, H = a.H;
  const size_t rowt = (size_t)b * S + t;
  (void)rho;
  double acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
  for (int j0 = 0; j0 < H; j0 += 16) {
    // A tile: 64 locations x 16 j  (1024 elements, 4 per thread)
    for (int e = tid; e < 1024; e += 256) {
      const int jj = e & 15, ll = e >> 4;
      const int l = l0 + ll, j = j0 + jj;
      double v = 0.0;
      if (l < L && j < H) {
        const size_t o = ((size_t)b * L + l) * H + j;
        const double vf = (double)a.vfeat[o] * (double)a.att[rowt * L + l];
        const float rV = (float)(vf * a.rctx[(size_t)n * H + j]);           // float32 store into r_V (E:554, :648)
        v = (double)rV * a.ipre[o];
      }
      As[jj][ll] = v;
    }
    for (int e = tid; e < 1024; e += 256) {
      const int dd = e & 63, jj = e >> 6;
      const int d = d0 + dd, j = j0 + jj;
      Bs[jj][dd] = (d < D && j < H) ? (double)a.WifT[(size_t)j * D + d] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
      double av[4], bv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) av[i] = As[jj][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) bv[j] = Bs[jj][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += av[i] * bv[j];
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int l = l0 + ty * 4 + i;
    if (l >= L) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int d = d0 + tx * 4 + j;
      if (d >= D) continue;
      const float f = a.F[((size_t)b * L + l) * D + d];
      const float fl = f / (float)L;                                   // float32 division, as numpy does
      const float first = (float)((double)fl / stab((double)a.avg[(size_t)b * D + d]) * a.ravg[(size_t)n * D + d]);
      a.R_feat[((size_t)n * L + l) * D + d] = (float)((double)first + (double)f * acc[i][j]);
    }
  }
}

// ------------------------------------------------------------------------------------------
// Tail on the matrix cores (replaces dec_tail_kernel's float64 VALU GEMM): the (L x H).(H x D) product per
// token runs on conv_igemm (1-tap mode) with the A operand materialised once by tail_a_kernel and the
// F-multiply fused as the EPI_MUL gate; tail_finish_kernel adds the mean-pool share.  The reference rounds
// r_V and R_feat to float32 anyway (E:554-555); the GEMM itself now rounds at fp32 / split-bf16 level
// (~1e-6 relative on R_feat, bar 1e-4).
//   A[n][l][j] = float32( float32( relu(if_pre[l][j]) * alpha_l * rho_j ) / stab(if_pre[l][j]) )
// ------------------------------------------------------------------------------------------
struct TailAArgs {
  const int* img_idx; const int* tpos;
  const float* vfeat; const double* ipre; const float* att; const double* rho;
  float* A;                 // [n][L][H] fp32, or split8 when `split`
  int Tm, L, H, split;
};

__global__ __launch_bounds__(256) void tail_a_kernel(TailAArgs a) {
  const int n = blockIdx.y;
  const int b = a.img_idx[n], t = a.tpos[n], S = a.Tm + 1;
  const int H8 = a.H >> 3;
  const size_t rowt = (size_t)b * S + t;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < a.L * H8; i += gridDim.x * 256) {
    const int l = i / H8, j0 = (i - l * H8) << 3;
    const size_t o = ((size_t)b * a.L + l) * a.H + j0;
    const double al = (double)a.att[rowt * a.L + l];
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float rV = (float)((double)a.vfeat[o + q] * al * a.rho[(size_t)n * a.H + j0 + q]);
      v[q] = (float)((double)rV * a.ipre[o + q]);
    }
    float* dst = a.A + ((size_t)n * a.L + l) * a.H + j0;
    if (a.split) {
      split8_store(v, dst);
    } else {
      *reinterpret_cast<f32x4*>(dst) = *reinterpret_cast<const f32x4*>(v);
      *reinterpret_cast<f32x4*>(dst + 4) = *reinterpret_cast<const f32x4*>(v + 4);
    }
  }
}

// R_feat[n][l][d] = float32( float32( F/L / stab(avg) * r_avg ) + R_feat[n][l][d] )      (E:642-647 + E:654-659)
__global__ __launch_bounds__(256) void tail_finish_kernel(const int* __restrict__ img_idx, const float* __restrict__ F,
                                                          const float* __restrict__ avg, const double* __restrict__ ravg,
                                                          float* __restrict__ R, int L, int D) {
  const int n = blockIdx.y, b = img_idx[n];
  for (int i = blockIdx.x * 256 + threadIdx.x; i < L * D; i += gridDim.x * 256) {
    const int d = i % D;
    const float f = F[(size_t)b * L * D + i];
    const float fl = f / (float)L;
    const float first = (float)((double)fl / stab((double)avg[(size_t)b * D + d]) * ravg[(size_t)n * D + d]);
    float* r = R + (size_t)n * L * D + i;
    *r = (float)((double)first + (double)*r);
  }
}

}  // namespace lrp
