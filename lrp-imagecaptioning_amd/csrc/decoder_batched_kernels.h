// decoder_batched_kernels.h — the per-token decoder LRP (E:537-666, E:1180-1321) run STEP-SYNCHRONOUSLY over all
// (image, token) units of a call: at scan step s every unit handles its LSTM step i = t-1-s, so the gate-g rule's
// product with the (2E+H) x H weight block is ONE (n x H).(H x Nd) GEMM on the matrix cores per step (conv_igemm,
// 1 tap, exact fp32) instead of one GEMV per unit that re-reads 3 MB of weights from L2 (dec_explain_adaptive_kernel /
// gtd_explain_kernel: one workgroup per unit, kept as the fallback for H % 4 != 0).  The element-wise rule arithmetic
// stays in float64 exactly as in those kernels; only the GEMM operand q = r_g / stab(g) and its result pass through
// float32 (6e-8 relative, bar 1e-4).
#pragma once
#include "decoder_kernels.h"

namespace lrp {

struct BxArgs {                                    // adaptive
  const int* img_idx; const int* tpos; const int* cap;
  const float *ht, *ct, *gt, *it, *ft, *st, *beta, *att, *xt;
  const double *ctx, *chat, *preds;
  const float* Wout; const float* WglobT; const float *avg, *glob_pre;
  double *rc, *rh, *rglob;                         // [n][H], [n][H], [n][E]   scan state
  float* q32;                                      // [n][H]      GEMM operand
  const float* acc32;                              // [n][2E+H]   GEMM result
  double *rctx, *ravg;
  float* att_out; double* rwords_out;
  int Tm, L, D, H, E, V, single_step;
};

// head (E:552-602), identical arithmetic to dec_explain_adaptive_kernel
__global__ __launch_bounds__(256) void bx_head_kernel(BxArgs a) {
  const int H = a.H, E = a.E, Tm = a.Tm, S = Tm + 1;
  const int n = blockIdx.x, tid = threadIdx.x;
  const int b = a.img_idx[n], t = a.tpos[n];
  const size_t rowt = (size_t)b * S + t;
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
    a.rctx[(size_t)n * H + j] = r_ctx / stab(a.ctx[rowt * H + j]);
    a.rc[(size_t)n * H + j] = r_s;
    a.rh[(size_t)n * H + j] = r_h;
  }
  for (int e = tid; e < E; e += 256) a.rglob[(size_t)n * E + e] = 0.0;
  if (a.att_out)
    for (int l = tid; l < a.L; l += 256) a.att_out[(size_t)n * a.L + l] = a.att[rowt * a.L + l];
  if (a.rwords_out)
    for (int i = tid; i < Tm; i += 256) a.rwords_out[(size_t)n * Tm + i] = 0.0;
}

// cell rules of scan step s (E:604-619) -> q; units that are done write a zero row
__device__ __forceinline__ void bx_pre_body(const BxArgs& a, int s) {
  const int H = a.H, S = a.Tm + 1;
  const int n = blockIdx.x, tid = threadIdx.x;
  const int b = a.img_idx[n], i = a.tpos[n] - 1 - s;
  float* q = a.q32 + (size_t)n * H;
  if (i < 0 || (a.single_step && s > 0)) {
    for (int j = tid; j < H; j += 256) q[j] = 0.f;
    return;
  }
  const size_t r1 = ((size_t)b * S + i + 1) * H, r0 = ((size_t)b * S + i) * H;
  double* rc = a.rc + (size_t)n * H;
  const double* rh = a.rh + (size_t)n * H;
  for (int j = tid; j < H; j += 256) {
    const double rcj = rc[j] + rh[j];
    const double sc = stab((double)a.ct[r1 + j]);
    const float pg = a.it[r1 + j] * tanhf(a.gt[r1 + j]);
    const float pc = a.ft[r1 + j] * a.ct[r0 + j];
    const double r_g = (double)pg / sc * rcj;
    rc[j] = (double)pc / sc * rcj;
    q[j] = (float)(r_g / stab((double)a.gt[r1 + j]));
  }
}
__global__ __launch_bounds__(256) void bx_pre_kernel(BxArgs a, int s) { bx_pre_body(a, s); }

// input rule of scan step s (E:620-632): r_x = x * (W_g q), routed to r_words / r_glob / r_h — and, next != 0, the cell rules
// of step s + 1 behind it in the same launch (a unit is one workgroup in both: the barrier between them orders r_h)
__global__ __launch_bounds__(256) void bx_post_kernel(BxArgs a, int s, int next) {
  __shared__ double red[4];
  const int H = a.H, E = a.E, Tm = a.Tm, S = Tm + 1, Nd = 2 * E + H;
  const int n = blockIdx.x, tid = threadIdx.x;
  const int b = a.img_idx[n], i = a.tpos[n] - 1 - s;
  if (!(i < 0 || (a.single_step && s > 0))) {            // (block-uniform)
    const size_t r0 = ((size_t)b * S + i) * H;
    const float* acc = a.acc32 + (size_t)n * Nd;
    double wsum = 0.0;
    for (int d = tid; d < Nd; d += 256) {
      const float x = d < 2 * E ? a.xt[((size_t)b * Tm + i) * 2 * E + d] : a.ht[r0 + d - 2 * E];
      const double rx = (double)x * (double)acc[d];
      if (d < E) wsum += rx;
      else if (d < 2 * E) a.rglob[(size_t)n * E + d - E] += rx;
      else a.rh[(size_t)n * H + d - 2 * E] = rx;
    }
    const double ws = block_sum_d(wsum, red);
    if (tid == 0 && a.rwords_out) a.rwords_out[(size_t)n * Tm + i] = ws;
  }
  if (next) {
    __syncthreads();                                     // this unit's r_h of step s is complete (global writes of the workgroup)
    bx_pre_body(a, s + 1);
  }
}

// ravg[d] = avg[d] * sum_e WglobT[e][d] q[e]  (float64, e in order)
__device__ __forceinline__ void bx_tail_column(const float* __restrict__ WglobT, const double* q, const float* __restrict__ avg,
                                               double* __restrict__ ravg, int d, int E, int D) {
  if (d >= D) return;
  const float* w = WglobT + d;
  double s = 0.0;
  constexpr int U = 16;
  int e = 0;
  for (; e + U <= E; e += U) {
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = w[(size_t)(e + u) * D];
#pragma unroll
    for (int u = 0; u < U; ++u) s += (double)v[u] * q[e + u];
  }
  for (; e < E; ++e) s += (double)w[(size_t)e * D] * q[e];
  ravg[d] = (double)avg[d] * s;
}

// global-feature rule (E:634-639) + r_words post-processing (E:660-665)
__global__ __launch_bounds__(256) void bx_tail_kernel(BxArgs a) {
  extern __shared__ double dsm[];
  double* q = dsm;
  const int E = a.E, D = a.D, Tm = a.Tm;
  // grid (n, ceil(D / 64)), 64 threads: one column d per thread, the E products added in order, 16 weight loads at a time
  // (one 256-thread block per token walking E rows for two columns per thread: 110 us for ten tokens [MI355X])
  const int n = blockIdx.x, tid = threadIdx.x;
  const int b = a.img_idx[n], t = a.tpos[n];
  for (int e = tid; e < E; e += 64) q[e] = a.rglob[(size_t)n * E + e] / stab((double)a.glob_pre[(size_t)b * E + e]);
  __syncthreads();
  bx_tail_column(a.WglobT, q, a.avg + (size_t)b * D, a.ravg + (size_t)n * D, blockIdx.y * 64 + tid, E, D);
  if (a.rwords_out && !a.single_step && tid == 0 && blockIdx.y == 0) {
    double* rw = a.rwords_out + (size_t)n * Tm;
    rw[0] = 0.0;
    double m = 0.0;
    for (int i = 0; i < t; ++i) m = fmax(m, fabs(rw[i]));
    for (int i = 0; i + 1 < t; ++i) rw[i] = m != 0.0 ? rw[i + 1] / m : rw[i + 1];
    rw[t - 1] = 0.0;
  }
}

// ------------------------------------------------------------------------------------------------------------
// grid-TD (E:1180-1321): the same step-synchronous scan with two LSTMs per step — language cell -> GEMM (3H columns)
// -> routing + c_hat split + top-down cell -> GEMM (H+2E+H columns) -> routing.  Arithmetic transcribed from
// gtd_explain_kernel (decoder_gridtd_kernels.h), state in float64 per unit instead of LDS.
// ------------------------------------------------------------------------------------------------------------
struct GbxArgs {
  const int* img_idx; const int* tpos; const int* cap;
  const double *h1t, *c1t, *g1t, *i1t, *f1t, *h2t, *c2t, *g2t, *i2t, *f2t, *x1t, *x2t;
  const double *ctx, *st, *chat, *beta, *att, *preds;
  const float* Wout; const float* WglobT; const float *avg, *glob_pre;
  double *rc1, *rc2, *rh1, *rh2, *rchat, *nh1, *nh2, *rglob;      // [n][H] each, rglob [n][E]
  float* q32; const float* acc32;
  double *rho, *ravg;
  float* att_out; double* rwords_out;
  int Tm, L, D, H, E, V;
};

__global__ __launch_bounds__(256) void gbx_head_kernel(GbxArgs a) {                  // E:1212-1229
  const int H = a.H, E = a.E, Tm = a.Tm, S = Tm + 1;
  const int n = blockIdx.x, tid = threadIdx.x;
  const int b = a.img_idx[n], t = a.tpos[n];
  const size_t rowt = (size_t)b * S + t, o = (size_t)n * H;
  const int k = a.cap[b * Tm + t - 1] - 1;
  const double zk = a.preds[((size_t)b * Tm + (t - 1)) * a.V + k];
  for (int j = tid; j < H; j += 256) {
    const double h2 = a.h2t[rowt * H + j], ch = a.chat[rowt * H + j];
    const double u = h2 + ch;
    const double r_u = ((double)a.Wout[(size_t)j * a.V + k] * u) / stab(zk) * zk;
    a.rh2[o + j] = h2 / stab(u) * r_u;
    a.rchat[o + j] = ch / stab(u) * r_u;
    a.rc1[o + j] = 0.0; a.rc2[o + j] = 0.0; a.rh1[o + j] = 0.0;
  }
  for (int e = tid; e < E; e += 256) a.rglob[(size_t)n * E + e] = 0.0;
  if (a.att_out)
    for (int l = tid; l < a.L; l += 256) a.att_out[(size_t)n * a.L + l] = (float)a.att[rowt * a.L + l];
  if (a.rwords_out)
    for (int i = tid; i < Tm; i += 256) a.rwords_out[(size_t)n * Tm + i] = 0.0;
}

__global__ __launch_bounds__(256) void gbx_pre2_kernel(GbxArgs a, int s) {           // language LSTM cell, E:1233-1251
  const int H = a.H, S = a.Tm + 1;
  const int n = blockIdx.x, tid = threadIdx.x;
  const int b = a.img_idx[n], i = a.tpos[n] - 1 - s;
  float* q = a.q32 + (size_t)n * H;
  if (i < 0) {
    for (int j = tid; j < H; j += 256) q[j] = 0.f;
    return;
  }
  const size_t r1 = ((size_t)b * S + i + 1) * H, r0 = ((size_t)b * S + i) * H, o = (size_t)n * H;
  for (int j = tid; j < H; j += 256) {
    const double rc = a.rc2[o + j] + a.rh2[o + j];
    const double sc = stab(a.c2t[r1 + j]);
    const double r_g = (a.i2t[r1 + j] * tanh(a.g2t[r1 + j])) / sc * rc;
    a.rc2[o + j] = (a.f2t[r1 + j] * a.c2t[r0 + j]) / sc * rc;
    q[j] = (float)(r_g / stab(a.g2t[r1 + j]));
  }
}

// routing of the language LSTM's input relevance (E:1252-1254), c_hat split (E:1255-1266), top-down cell (E:1268-1281)
__global__ __launch_bounds__(256) void gbx_mid_kernel(GbxArgs a, int s) {
  const int H = a.H, Tm = a.Tm, S = Tm + 1, Nd2 = 3 * H;
  const int n = blockIdx.x, tid = threadIdx.x;
  const int b = a.img_idx[n], t = a.tpos[n], i = t - 1 - s;
  float* q = a.q32 + (size_t)n * H;
  if (i < 0) {
    for (int j = tid; j < H; j += 256) q[j] = 0.f;
    return;
  }
  const size_t r1 = ((size_t)b * S + i + 1) * H, r0 = ((size_t)b * S + i) * H, o = (size_t)n * H;
  const float* acc = a.acc32 + (size_t)n * Nd2;
  for (int d = tid; d < Nd2; d += 256) {
    const double x = d < 2 * H ? a.x2t[((size_t)b * Tm + i) * 2 * H + d] : a.h2t[r0 + d - 2 * H];
    const double rx = x * (double)acc[d];
    if (d < H) a.rchat[o + d] = (i == t - 1 ? a.rchat[o + d] : 0.0) + rx;
    else if (d < 2 * H) a.rh1[o + d - H] += rx;
    else a.nh2[o + d - 2 * H] = rx;
  }
  __syncthreads();                                   // (one workgroup per unit: its own writes above are all it needs)
  const double bt = a.beta[(size_t)b * S + i + 1];
  for (int j = tid; j < H; j += 256) {
    const double sch = stab(a.chat[r1 + j]);
    const double r_s = (bt * a.st[r1 + j]) / sch * a.rchat[o + j];
    const double r_ctx = (a.ctx[r1 + j] * (1.0 - bt)) / sch * a.rchat[o + j];
    a.rho[((size_t)n * Tm + i) * H + j] = r_ctx / stab(a.ctx[r1 + j]);
    const double rc = (a.rc1[o + j] + r_s) + a.rh1[o + j];
    const double sc = stab(a.c1t[r1 + j]);
    const double r_g = (a.i1t[r1 + j] * tanh(a.g1t[r1 + j])) / sc * rc;
    a.rc1[o + j] = (a.f1t[r1 + j] * a.c1t[r0 + j]) / sc * rc;
    q[j] = (float)(r_g / stab(a.g1t[r1 + j]));
  }
}

__global__ __launch_bounds__(256) void gbx_post_kernel(GbxArgs a, int s) {           // E:1282-1300
  __shared__ double red[4];
  const int H = a.H, E = a.E, Tm = a.Tm, S = Tm + 1, K1 = H + 2 * E, Nd1 = K1 + H;
  const int n = blockIdx.x, tid = threadIdx.x;
  const int b = a.img_idx[n], i = a.tpos[n] - 1 - s;
  if (i < 0) return;
  const size_t r0 = ((size_t)b * S + i) * H, o = (size_t)n * H;
  const float* acc = a.acc32 + (size_t)n * Nd1;
  double wsum = 0.0;
  for (int d = tid; d < Nd1; d += 256) {
    const double x = d < K1 ? a.x1t[((size_t)b * Tm + i) * K1 + d] : a.h1t[r0 + d - K1];
    const double rx = x * (double)acc[d];
    if (d < H) a.nh2[o + d] += rx;
    else if (d < H + E) a.rglob[(size_t)n * E + d - H] += rx;
    else if (d < K1) wsum += rx;
    else a.nh1[o + d - K1] = rx;
  }
  const double ws = block_sum_d(wsum, red);          // (its barriers also order the nh1 / nh2 writes before the copies)
  if (tid == 0 && a.rwords_out) a.rwords_out[(size_t)n * Tm + i] = ws;
  for (int j = tid; j < H; j += 256) { a.rh2[o + j] = a.nh2[o + j]; a.rh1[o + j] = a.nh1[o + j]; }
}

__global__ __launch_bounds__(256) void gbx_tail_kernel(GbxArgs a) {                  // E:1301-1306
  extern __shared__ double dsm[];
  double* q = dsm;
  const int E = a.E, D = a.D;
  const int n = blockIdx.x, tid = threadIdx.x, b = a.img_idx[n];
  for (int e = tid; e < E; e += 64) q[e] = a.rglob[(size_t)n * E + e] / stab((double)a.glob_pre[(size_t)b * E + e]);
  __syncthreads();
  bx_tail_column(a.WglobT, q, a.avg + (size_t)b * D, a.ravg + (size_t)n * D, blockIdx.y * 64 + tid, E, D);
}

}  // namespace lrp
