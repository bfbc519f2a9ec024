// gradient_kernels.h — the reference's gradient BASELINES: the hand-written BPTT of
// ExplainImgCaptioning{AdaptiveAttention,GridTD}Gradient._lstm_decoder_backward (E:780-832, E:1452-1532),
// batched over all (image, token) units of a call and run step-synchronously: at scan step s every unit
// processes its LSTM step i = t-1-s, so the two transposed-weight products of a step are ONE GEMM over all units
// (M = n) on the MFMA kernel (conv_igemm, 1 tap, exact fp32) instead of n GEMVs that each re-read 12 MB of weights.
// The simplifications of the reference's backward (attention / beta / sentinel treated as constants, see
// oracle/decoder_grad_ref.py) are part of the specification and are kept.
#pragma once
#include <hip/hip_runtime.h>

namespace lrp {

// seed[u][h] = W_out[h][k_u]  (E:801 / E:1484: one-hot . W_out^T), k_u = caption[b][t-1] - 1; clears the per-call accumulators
__global__ __launch_bounds__(256) void grad_seed_kernel(const int* __restrict__ img_idx, const int* __restrict__ tpos,
                                                        const int* __restrict__ cap, const float* __restrict__ Wout,
                                                        float* __restrict__ seed, float* __restrict__ dc1,
                                                        float* __restrict__ dc2, float* __restrict__ dglob,
                                                        double* __restrict__ dwords, int Tm, int H, int E, int V) {
  const int u = blockIdx.x, b = img_idx[u], t = tpos[u];
  const int k = cap[(size_t)b * Tm + t - 1] - 1;
  for (int j = threadIdx.x; j < H; j += 256) {
    seed[(size_t)u * H + j] = (k >= 0 && k < V) ? Wout[(size_t)j * V + k] : 0.f;
    dc1[(size_t)u * H + j] = 0.f;
    if (dc2) dc2[(size_t)u * H + j] = 0.f;
  }
  for (int j = threadIdx.x; j < E; j += 256) dglob[(size_t)u * E + j] = 0.f;
  for (int j = threadIdx.x; j < Tm; j += 256) dwords[(size_t)u * Tm + j] = 0.0;
}

// One LSTM cell backward (E:811-822 / E:1490-1500, E:1505-1515) for every unit at scan step s (i = t-1-s):
//   dh = [s == 0] seed + [s > 0] A[u][offA..] + B[u][offB..]   (A: carried over the previous step, B: produced this step)
//   dc' = dc + dh o (1 - tanh^2 c);  d_f = dc' c_prev f(1-f);  d_i = dc' g i(1-i);  d_g = dc' i (1-g^2);  d_o = dh tanh(c) o(1-o)
//   dc <- dc' f;   dg[u] = [d_i | d_f | d_g | d_o]  (zeros once i < 0)
// TS = float (adaptive state arrays) or double (grid-TD).  gt holds the PRE-activation of g (E:134), g = tanh(gt).
template <typename TS>
__global__ __launch_bounds__(256) void grad_cell_kernel(const int* __restrict__ img_idx, const int* __restrict__ tpos, int s,
                                                        const float* __restrict__ seed, const float* __restrict__ A, int ldA,
                                                        int offA, const float* __restrict__ Bv, int ldB, int offB,
                                                        const TS* __restrict__ ct, const TS* __restrict__ it,
                                                        const TS* __restrict__ ft, const TS* __restrict__ gt,
                                                        const TS* __restrict__ ot, float* __restrict__ dc,
                                                        float* __restrict__ dg, int Tm, int H) {
  const int u = blockIdx.x, b = img_idx[u], i = tpos[u] - 1 - s;
  float* dgu = dg + (size_t)u * 4 * H;
  if (i < 0) {
    for (int j = threadIdx.x; j < 4 * H; j += 256) dgu[j] = 0.f;
    return;
  }
  const size_t cur = ((size_t)b * (Tm + 1) + i + 1) * H, prev = cur - H;
  for (int j = threadIdx.x; j < H; j += 256) {
    float dh = 0.f;
    if (s == 0 && seed) dh = seed[(size_t)u * H + j];
    if (s > 0 && A) dh += A[(size_t)u * ldA + offA + j];
    if (Bv) dh += Bv[(size_t)u * ldB + offB + j];
    const float c = (float)ct[cur + j], cp = (float)ct[prev + j];
    const float ia = (float)it[cur + j], fa = (float)ft[cur + j], oa = (float)ot[cur + j], ga = tanhf((float)gt[cur + j]);
    const float tc = tanhf(c);
    const float dcv = dc[(size_t)u * H + j] + dh * oa * (1.f - tc * tc);
    dc[(size_t)u * H + j] = dcv * fa;
    dgu[j] = dcv * ga * ia * (1.f - ia);
    dgu[H + j] = dcv * cp * fa * (1.f - fa);
    dgu[2 * H + j] = dcv * ia * (1.f - ga * ga);
    dgu[3 * H + j] = dh * tc * oa * (1.f - oa);
  }
}

// After the input-side GEMM of step s: d_glob += d_x[offG..offG+E)  and  r_words[i] = sum_e d_x[offW..offW+E)
// (E:825-826 / E:1518-1519) for the units that were active at this step.
__global__ __launch_bounds__(256) void grad_accum_kernel(const int* __restrict__ tpos, int s, const float* __restrict__ X,
                                                         int ldX, int offG, int offW, float* __restrict__ dglob,
                                                         double* __restrict__ dwords, int Tm, int E) {
  const int u = blockIdx.x, i = tpos[u] - 1 - s;
  if (i < 0) return;
  const float* x = X + (size_t)u * ldX;
  __shared__ double red[256];
  double acc = 0.0;
  for (int j = threadIdx.x; j < E; j += 256) {
    dglob[(size_t)u * E + j] += x[offG + j];
    acc += (double)x[offW + j];
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) dwords[(size_t)u * Tm + i] = red[0];
}

// grid-TD, between the two cells of a step (E:1502-1503): d_context_hat[i] = [s == 0] seed + d_x2[:H];
// dctx[u][i] = d_context_hat[i] (1 - beta[i+1])   — kept per step for the attention-sum of the tail.
__global__ __launch_bounds__(256) void gtd_grad_ctx_kernel(const int* __restrict__ img_idx, const int* __restrict__ tpos, int s,
                                                           const float* __restrict__ seed, const float* __restrict__ X2,
                                                           int ldX, int off, const double* __restrict__ beta,
                                                           float* __restrict__ dctx, int Tm, int H) {
  const int u = blockIdx.x, b = img_idx[u], i = tpos[u] - 1 - s;
  if (i < 0) return;
  const float ob = 1.f - (float)beta[(size_t)b * (Tm + 1) + i + 1];
  for (int j = threadIdx.x; j < H; j += 256) {
    float v = X2[(size_t)u * ldX + off + j];
    if (s == 0) v += seed[(size_t)u * H + j];
    dctx[((size_t)u * Tm + i) * H + j] = v * ob;
  }
}

// d_glob relu mask (E:827: scalar quirk of the adaptive class — everything is zeroed iff the FIRST element of the
// global feature is not positive; E:1523: element-wise for grid-TD)
__global__ __launch_bounds__(256) void grad_glob_mask_kernel(const int* __restrict__ img_idx, const float* __restrict__ glob_pre,
                                                             float* __restrict__ dglob, int E, int scalar_quirk) {
  const int u = blockIdx.x, b = img_idx[u];
  const float* g = glob_pre + (size_t)b * E;
  const bool all_off = scalar_quirk && !(g[0] > 0.f);
  for (int j = threadIdx.x; j < E; j += 256)
    if (all_off || (!scalar_quirk && !(g[j] > 0.f))) dglob[(size_t)u * E + j] = 0.f;
}

// A operand of the tail GEMM, rows (u, l): d_V[l][h] masked by image_features > 0 (E:807-809 / E:1520-1521, E:1525)
//   adaptive: d_V = seed[h] * alpha[t][l];   grid-TD: d_V = sum_{i<t} dctx[i][h] * alpha[i+1][l]
template <typename TA>
__global__ __launch_bounds__(256) void grad_tail_a_kernel(const int* __restrict__ img_idx, const int* __restrict__ tpos,
                                                          const float* __restrict__ seed, const float* __restrict__ dctx,
                                                          const TA* __restrict__ att, const float* __restrict__ if_pre,
                                                          float* __restrict__ Aout, int Tm, int L, int H) {
  const int u = blockIdx.y, b = img_idx[u], t = tpos[u];
  const TA* ab = att + (size_t)b * (Tm + 1) * L;
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < L * H; idx += gridDim.x * 256) {
    const int l = idx / H, h = idx - l * H;
    float v = 0.f;
    if (if_pre[((size_t)b * L + l) * H + h] > 0.f) {
      if (dctx) {
        for (int i = 0; i < t; ++i) v += dctx[((size_t)u * Tm + i) * H + h] * (float)ab[(size_t)(i + 1) * L + l];
      } else {
        v = seed[(size_t)u * H + h] * (float)ab[(size_t)t * L + l];
      }
    }
    Aout[((size_t)u * L + l) * H + h] = v;
  }
}

// d_feat[u][l][d] += d_avg[u][d] / L   (E:829-831 / E:1527-1529)
__global__ __launch_bounds__(256) void grad_tail_finish_kernel(const float* __restrict__ davg, float* __restrict__ out, int L, int D) {
  const int u = blockIdx.y;
  const float inv = 1.0f / (float)L;
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < L * D; idx += gridDim.x * 256)
    out[(size_t)u * L * D + idx] += davg[(size_t)u * D + (idx % D)] * inv;
}

}  // namespace lrp
