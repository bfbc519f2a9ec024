// decoder_gridtd_kernels.h — grid-TD (bottom-up / top-down) decoder, E:995-1321.
//   forward replay   _forward_beam_search                 E:1092-1178
//   per-token LRP    _explain_lstm_single_word_sequence   E:1180-1321
// In the reference this decoder runs in float64 from the first step on (x2t is float64 because
// context_hat is, and np.vstack then promotes every state array), so everything here is double.
// Reference quirks reproduced on purpose: logits are cached from h2 alone (E:1154) while the
// output rule is fed h2 + c_hat (E:1212-1217); relevance routing uses '+=' (E:1252-1254, :1288,
// :1300); r_V is a float32 accumulator that is rounded after every step's '+=' (E:1189, :1293);
// r_words is not normalised (E:1320).
#pragma once
#include <hip/hip_runtime.h>
#include "decoder_kernels.h"

namespace lrp {

__device__ __forceinline__ double sigmoid_d(double x) { return 1.0 / (1.0 + exp(-x)); }

// xh1[b] = [ h2_{i} | relu(glob_pre) | emb(tok) | h1_{i} ]  (E:1131-1136) and x1t[b][i] = first H+2E
__global__ __launch_bounds__(256) void gtd_prep_x1_kernel(const float* __restrict__ emb, const float* __restrict__ glob_pre,
                                                          const double* __restrict__ h1t, const double* __restrict__ h2t,
                                                          const int* __restrict__ cap, double* __restrict__ xh1,
                                                          double* __restrict__ x1t, int step, int Tm, int E, int H, int V,
                                                          int sos) {
  const int b = blockIdx.x, S = Tm + 1;
  int tok = (step == 0 ? sos : cap[b * Tm + step - 1]) - 1;
  tok = tok < 0 ? 0 : (tok >= V ? V - 1 : tok);
  const int K1 = H + 2 * E, Nd = K1 + H;
  for (int e = threadIdx.x; e < Nd; e += 256) {
    double v;
    if (e < H) v = h2t[((size_t)b * S + step) * H + e];
    else if (e < H + E) v = (double)fmaxf(glob_pre[(size_t)b * E + e - H], 0.f);
    else if (e < K1) v = (double)emb[(size_t)tok * E + e - H - E];
    else v = h1t[((size_t)b * S + step) * H + e - K1];
    xh1[(size_t)b * Nd + e] = v;
    if (e < K1) x1t[((size_t)b * Tm + step) * K1 + e] = v;
  }
}

// LSTM pointwise (E:129-138), optionally with the sentinel s = tanh(c) * sigmoid(gate)  (E:1145)
// z arrives as `ks` split-K slabs (`slab` doubles apart), added here in a fixed order
__global__ __launch_bounds__(256) void gtd_pointwise_kernel(const double* __restrict__ z, int ldz, int ks, size_t slab,
                                                            double* __restrict__ ht,
                                                            double* __restrict__ ct, double* __restrict__ gt,
                                                            double* __restrict__ it, double* __restrict__ ft,
                                                            double* __restrict__ st, double* __restrict__ hu,
                                                            double* __restrict__ ot, int step, int Tm, int H) {
  const int b = blockIdx.x, S = Tm + 1;
  const double* zb = z + (size_t)b * ldz;
  const size_t prev = ((size_t)b * S + step) * H, cur = prev + H;
  for (int j = threadIdx.x; j < H; j += 256) {
    double zz[5];
    const int ng = st ? 5 : 4;
    for (int g = 0; g < ng; ++g) {
      double v = zb[g * H + j];
      for (int q = 1; q < ks; ++q) v += zb[(size_t)q * slab + g * H + j];
      zz[g] = v;
    }
    const double i_ = sigmoid_d(zz[0]), f_ = sigmoid_d(zz[1]), g_ = zz[2], o_ = sigmoid_d(zz[3]);
    const double c = f_ * ct[prev + j] + i_ * tanh(g_);
    const double tc = tanh(c);
    const double h = o_ * tc;
    ht[cur + j] = h;
    ct[cur + j] = c;
    gt[cur + j] = g_;
    it[cur + j] = i_;
    ft[cur + j] = f_;
    ot[cur + j] = o_;                                  // (gradient baselines, E:1327-1342)
    if (st) st[cur + j] = tc * sigmoid_d(zz[4]);
    if (hu) hu[((size_t)b * Tm + step) * H + j] = h;          // rows of the output-layer GEMM (h2 only, E:1154)
  }
}

// attention + sentinel mix on h1 (E:1140-1149), then xh2[b] = [ c_hat | h1 | h2_{i} ] and x2t (E:1151)
// dynamic LDS: double hp[H], sp[H], pre[L+1]
__global__ __launch_bounds__(256) void gtd_attention_kernel(const double* __restrict__ hproj, const double* __restrict__ sproj,
                                                            int ks, size_t slab, const float* __restrict__ proj, const float* __restrict__ wa,
                                                            const float* __restrict__ if_pre, const double* __restrict__ h1t,
                                                            const double* __restrict__ h2t, const double* __restrict__ st,
                                                            double* __restrict__ att, double* __restrict__ beta,
                                                            double* __restrict__ ctx, double* __restrict__ chat,
                                                            double* __restrict__ xh2, double* __restrict__ x2t, int step,
                                                            int Tm, int L, int H) {
  extern __shared__ double gsm[];
  double* hp = gsm;
  double* sp = gsm + H;
  double* pre = gsm + 2 * H;
  const int b = blockIdx.x, S = Tm + 1, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int j = tid; j < H; j += 256) {
    double hv = hproj[(size_t)b * H + j], sv = sproj[(size_t)b * H + j];
    for (int q = 1; q < ks; ++q) { hv += hproj[(size_t)q * slab + (size_t)b * H + j]; sv += sproj[(size_t)q * slab + (size_t)b * H + j]; }
    hp[j] = hv; sp[j] = sv;
  }
  __syncthreads();
  for (int l = wave; l <= L; l += 4) {
    double p = 0.0;
    if (l < L) {
      const float* prow = proj + ((size_t)b * L + l) * H;
      for (int j = lane; j < H; j += 64) p += tanh((double)prow[j] + hp[j]) * (double)wa[j];
    } else {
      for (int j = lane; j < H; j += 64) p += tanh(sp[j] + hp[j]) * (double)wa[j];
    }
    p = wave_sum_d(p);
    if (lane == 0) pre[l] = p;
  }
  __syncthreads();
  const size_t row = (size_t)b * S + step + 1;
  if (wave == 0) {
    double mx = -1e300;
    for (int l = lane; l < L; l += 64) mx = fmax(mx, pre[l]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
    double sm = 0.0;
    for (int l = lane; l < L; l += 64) sm += exp(pre[l] - mx);
    sm = wave_sum_d(sm);
    const double zt = pre[L], mx2 = fmax(mx, zt);
    double sm2 = 0.0;
    for (int l = lane; l < L; l += 64) sm2 += exp(pre[l] - mx2);
    sm2 = wave_sum_d(sm2);
    const double ez = exp(zt - mx2), bt = ez / (sm2 + ez);
    for (int l = lane; l < L; l += 64) {
      const double al = exp(pre[l] - mx) / sm;
      pre[l] = al;
      att[row * L + l] = al;
    }
    if (lane == 0) { pre[L] = bt; beta[row] = bt; }
  }
  __syncthreads();
  const double bt = pre[L];
  for (int j = tid; j < H; j += 256) {
    double c = 0.0;
    for (int l = 0; l < L; ++l) c += pre[l] * (double)fmaxf(if_pre[((size_t)b * L + l) * H + j], 0.f);
    const double ch = bt * st[row * H + j] + (1.0 - bt) * c;
    const double h1 = h1t[row * H + j];
    ctx[row * H + j] = c;
    chat[row * H + j] = ch;
    double* x = xh2 + (size_t)b * 3 * H;
    x[j] = ch;
    x[H + j] = h1;
    x[2 * H + j] = h2t[((size_t)b * S + step) * H + j];
    x2t[((size_t)b * Tm + step) * 2 * H + j] = ch;
    x2t[((size_t)b * Tm + step) * 2 * H + H + j] = h1;
  }
}

// ------------------------------------------------------------------------------------------
// Per-token LRP (closed form of E:1180-1321), one workgroup per (image, t).
// Outputs: rho[n][i][j] = r_context[i][j] / stab(context[i+1][j]) for the attention-sum rule that
// the tail accumulates over every step (E:1292-1299), ravg[n][D], r_words[n][0..t).
// dynamic LDS (doubles): rc1 rc2 rh1 rh2 rchat q[H] each, rglob[E], nh1[H] nh2[H], red[4]
// ------------------------------------------------------------------------------------------
struct GtdExplainArgs {
  const int* img_idx; const int* tpos; const int* cap;
  const double *h1t, *c1t, *g1t, *i1t, *f1t, *h2t, *c2t, *g2t, *i2t, *f2t, *x1t, *x2t;
  const double *ctx, *st, *chat, *beta, *att, *preds;
  const float* Wout;        // [H][V]
  const float* Wg1T;        // [H][H+2E+H]   top-down LSTM gate-g block, transposed
  const float* Wg2T;        // [H][3H]       language LSTM gate-g block, transposed
  const float* WglobT;      // [E][D]
  const float *avg, *glob_pre;
  double *rho, *ravg;       // [n][Tm][H], [n][D]
  float* att_out;           // [n][L] or null  (cast to float32 for the ABI; the cached state stays float64)
  double* rwords_out;       // [n][Tm] or null
  int Tm, L, D, H, E, V;
};

__device__ __forceinline__ void gtd_gemv(const float* __restrict__ WT, int Nd, int H, const double* q, double* acc) {
  const int tid = threadIdx.x;
  int dcl[SCAN_MAXR];
#pragma unroll
  for (int r = 0; r < SCAN_MAXR; ++r) { acc[r] = 0.0; const int d = tid + 256 * r; dcl[r] = d < Nd ? d : Nd - 1; }
  const int nr = (Nd + 255) >> 8;
  int j = 0;
  for (; j + 4 <= H; j += 4) {
    float wv[4][SCAN_MAXR];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int r = 0; r < SCAN_MAXR; ++r)
        if (r < nr) wv[u][r] = WT[(size_t)(j + u) * Nd + dcl[r]];
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
      if (r < nr) acc[r] += (double)WT[(size_t)j * Nd + dcl[r]] * qj;
  }
}

__global__ __launch_bounds__(256) void gtd_explain_kernel(GtdExplainArgs a) {
  extern __shared__ double dsm[];
  const int H = a.H, E = a.E, D = a.D, Tm = a.Tm, S = Tm + 1;
  double* rc1 = dsm;
  double* rc2 = rc1 + H;
  double* rh1 = rc2 + H;
  double* rh2 = rh1 + H;
  double* rchat = rh2 + H;
  double* q = rchat + H;
  double* nh1 = q + (H > E ? H : E);
  double* nh2 = nh1 + H;
  double* rglob = nh2 + H;
  double* red = rglob + E;
  const int n = blockIdx.x, tid = threadIdx.x;
  const int b = a.img_idx[n], t = a.tpos[n];
  const int K1 = H + 2 * E, Nd1 = K1 + H, Nd2 = 3 * H;
  const size_t rowt = (size_t)b * S + t;

  // ---- head (E:1212-1229): seed through u = h2 + c_hat, logits from h2 alone
  const int k = a.cap[b * Tm + t - 1] - 1;
  const double zk = a.preds[((size_t)b * Tm + (t - 1)) * a.V + k];
  for (int j = tid; j < H; j += 256) {
    const double h2 = a.h2t[rowt * H + j], ch = a.chat[rowt * H + j];
    const double u = h2 + ch;
    const double r_u = ((double)a.Wout[(size_t)j * a.V + k] * u) / stab(zk) * zk;
    rh2[j] = h2 / stab(u) * r_u;
    rchat[j] = ch / stab(u) * r_u;                 // r_context_hat[t-1]
    rc1[j] = 0.0; rc2[j] = 0.0; rh1[j] = 0.0;
  }
  for (int e = tid; e < E; e += 256) rglob[e] = 0.0;
  if (a.att_out)
    for (int l = tid; l < a.L; l += 256) a.att_out[(size_t)n * a.L + l] = (float)a.att[rowt * a.L + l];
  if (a.rwords_out)
    for (int i = tid; i < Tm; i += 256) a.rwords_out[(size_t)n * Tm + i] = 0.0;
  __syncthreads();

  for (int i = t - 1; i >= 0; --i) {
    const size_t r1 = ((size_t)b * S + i + 1) * H, r0 = ((size_t)b * S + i) * H;
    // ---- language LSTM (E:1233-1254)
    for (int j = tid; j < H; j += 256) {
      const double rc = rc2[j] + rh2[j];
      const double sc = stab(a.c2t[r1 + j]);
      const double r_g = (a.i2t[r1 + j] * tanh(a.g2t[r1 + j])) / sc * rc;
      rc2[j] = (a.f2t[r1 + j] * a.c2t[r0 + j]) / sc * rc;
      q[j] = r_g / stab(a.g2t[r1 + j]);
    }
    __syncthreads();
    double acc[SCAN_MAXR];
    gtd_gemv(a.Wg2T, Nd2, H, q, acc);
#pragma unroll
    for (int r = 0; r < SCAN_MAXR; ++r) {
      const int d = tid + 256 * r;
      if (d < Nd2) {
        const double x = d < 2 * H ? a.x2t[((size_t)b * Tm + i) * 2 * H + d] : a.h2t[r0 + d - 2 * H];
        const double rx = x * acc[r];
        if (d < H) rchat[d] = (i == t - 1 ? rchat[d] : 0.0) + rx;          // r_context_hat[i] +=
        else if (d < 2 * H) rh1[d - H] += rx;                              // r_h1t[i+1] +=
        else nh2[d - 2 * H] = rx;                                          // r_h2t[i] += (from zero)
      }
    }
    __syncthreads();
    // ---- split c_hat -> sentinel / context (E:1255-1266), top-down LSTM cell (E:1268-1281)
    const double bt = a.beta[(size_t)b * S + i + 1];
    for (int j = tid; j < H; j += 256) {
      const double sch = stab(a.chat[r1 + j]);
      const double r_s = (bt * a.st[r1 + j]) / sch * rchat[j];
      const double r_ctx = (a.ctx[r1 + j] * (1.0 - bt)) / sch * rchat[j];
      a.rho[((size_t)n * Tm + i) * H + j] = r_ctx / stab(a.ctx[r1 + j]);
      const double rc = (rc1[j] + r_s) + rh1[j];
      const double sc = stab(a.c1t[r1 + j]);
      const double r_g = (a.i1t[r1 + j] * tanh(a.g1t[r1 + j])) / sc * rc;
      rc1[j] = (a.f1t[r1 + j] * a.c1t[r0 + j]) / sc * rc;
      q[j] = r_g / stab(a.g1t[r1 + j]);
    }
    __syncthreads();
    gtd_gemv(a.Wg1T, Nd1, H, q, acc);
    double wsum = 0.0;
#pragma unroll
    for (int r = 0; r < SCAN_MAXR; ++r) {
      const int d = tid + 256 * r;
      if (d < Nd1) {
        const double x = d < K1 ? a.x1t[((size_t)b * Tm + i) * K1 + d] : a.h1t[r0 + d - K1];
        const double rx = x * acc[r];
        if (d < H) nh2[d] += rx;                                           // r_h2t[i] += r_xht1[:H]
        else if (d < H + E) rglob[d - H] += rx;
        else if (d < K1) wsum += rx;                                       // r_wordembedding[i]
        else nh1[d - K1] = rx;                                             // r_h1t[i] +=
      }
    }
    const double ws = block_sum_d(wsum, red);
    if (tid == 0 && a.rwords_out) a.rwords_out[(size_t)n * Tm + i] = ws;
    for (int j = tid; j < H; j += 256) { rh2[j] = nh2[j]; rh1[j] = nh1[j]; }
    __syncthreads();
  }

  // ---- global-feature rule (E:1301-1306)
  for (int e = tid; e < E; e += 256) q[e] = rglob[e] / stab((double)a.glob_pre[(size_t)b * E + e]);
  __syncthreads();
  for (int d = tid; d < D; d += 256) {
    double s = 0.0;
    for (int e = 0; e < E; ++e) s += (double)a.WglobT[(size_t)e * D + d] * q[e];
    a.ravg[(size_t)n * D + d] = (double)a.avg[(size_t)b * D + d] * s;
  }
}

// Tail (E:1307-1319) with the attention-sum rule accumulated over every step (E:1292-1299):
//   r_V[l][j] = fold_{i=t-1..0} float32( r_V + relu(if_pre[l][j]) * att[i+1][l] * rho[i][j] )
struct GtdTailArgs {
  const int* img_idx; const int* tpos;
  const float* F; const float* if_pre; const double* att; const float* avg; const float* WifT;
  const double *rho, *ravg;
  float* R_feat;
  int Tm, L, D, H;
};

// A operand of the same tail on the matrix cores (see tail_a_kernel in decoder_kernels.h): the per-step fold with its
// float32 rounding after every += (E:1189, E:1293) and the division by stab(pre) happen here, the (L x H).(H x D)
// product runs on conv_igemm with the F-multiply as its gate, tail_finish_kernel adds the mean-pool share.
//   A[n][l][j] = float32( r_V[l][j] / stab(if_pre[l][j]) ),  r_V folded over i = t-1 .. 0 as in gtd_tail_kernel
__global__ __launch_bounds__(256) void gtd_tail_a_kernel(const int* __restrict__ img_idx, const int* __restrict__ tpos,
                                                         const float* __restrict__ if_pre, const double* __restrict__ att,
                                                         const double* __restrict__ rho, float* __restrict__ A, int Tm,
                                                         int L, int H, int split) {
  const int n = blockIdx.y, b = img_idx[n], t = tpos[n], S = Tm + 1, H8 = H >> 3;
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < L * H8; idx += gridDim.x * 256) {
    const int l = idx / H8, j0 = (idx - l * H8) << 3;
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float pre = if_pre[((size_t)b * L + l) * H + j0 + q];
      const double vf = (double)fmaxf(pre, 0.f);
      float rV = 0.f;
      for (int i = t - 1; i >= 0; --i)
        rV = (float)((double)rV + vf * att[((size_t)b * S + i + 1) * L + l] * rho[((size_t)n * Tm + i) * H + j0 + q]);
      v[q] = (float)((double)rV / stab((double)pre));
    }
    float* dst = A + ((size_t)n * L + l) * H + j0;
    if (split) {
      split8_store(v, dst);
    } else {
      *reinterpret_cast<f32x4*>(dst) = *reinterpret_cast<const f32x4*>(v);
      *reinterpret_cast<f32x4*>(dst + 4) = *reinterpret_cast<const f32x4*>(v + 4);
    }
  }
}

__global__ __launch_bounds__(256) void gtd_tail_kernel(GtdTailArgs a) {
  __shared__ double As[16][65];
  __shared__ double Bs[16][65];
  const int n = blockIdx.x, l0 = blockIdx.y * 64, d0 = blockIdx.z * 64;
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int b = a.img_idx[n], t = a.tpos[n], S = a.Tm + 1;
  const int L = a.L, D = a.D, H = a.H;
  double acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
  for (int j0 = 0; j0 < H; j0 += 16) {
    for (int e = tid; e < 1024; e += 256) {
      const int jj = e & 15, ll = e >> 4;
      const int l = l0 + ll, j = j0 + jj;
      double v = 0.0;
      if (l < L && j < H) {
        const float pre = a.if_pre[((size_t)b * L + l) * H + j];
        const double vf = (double)fmaxf(pre, 0.f);
        float rV = 0.f;
        for (int i = t - 1; i >= 0; --i)
          rV = (float)((double)rV + vf * a.att[((size_t)b * S + i + 1) * L + l] * a.rho[((size_t)n * a.Tm + i) * H + j]);
        v = (double)rV / stab((double)pre);
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
      const float fl = f / (float)L;
      const float first = (float)((double)fl / stab((double)a.avg[(size_t)b * D + d]) * a.ravg[(size_t)n * D + d]);
      a.R_feat[((size_t)n * L + l) * D + d] = (float)((double)first + (double)f * acc[i][j]);
    }
  }
}

}  // namespace lrp
