// Point-wise / attention kernels of the fine-tune step of the adaptive-attention captioner.
// Forward graph: models/model.py:1340-1368 (build) + :573-600 (ExternalAttentionRNNWrapperLocalAttentionV3.step);
// loss :95-103 with loss_weights [0.5, 0.5] (:1370-1373).  Everything is fp32, time-major (t, b) rows.
#pragma once
#include <hip/hip_runtime.h>

namespace lrp {

__device__ __forceinline__ float tr_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

__device__ __forceinline__ float tr_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// sum over a 256-thread block, result broadcast to every thread (red: 4 floats of LDS + 1)
__device__ __forceinline__ float tr_block_sum(float v, float* red) {
  v = tr_wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  const float s = red[0] + red[1] + red[2] + red[3];
  return s;
}
__device__ __forceinline__ float tr_block_max(float v, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// x[r][n] = relu(x[r][n] + b[n]) * mask[r][n]   (Dense(relu) -> Dropout, M:1346-1352)
__global__ __launch_bounds__(256) void tr_bias_relu_mask_kernel(float* __restrict__ x, const float* __restrict__ b,
                                                                const float* __restrict__ mask, size_t rows, int N) {
  const size_t n = rows * N;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float v = fmaxf(x[i] + b[i % N], 0.f);
    if (mask) v *= mask[i];
    x[i] = v;
  }
}
// d[i] = d[i] * mask[i] * [y[i] > 0]   (back through Dropout and ReLU; y = the masked activation)
__global__ __launch_bounds__(256) void tr_relu_mask_bwd_kernel(float* __restrict__ d, const float* __restrict__ y,
                                                               const float* __restrict__ mask, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float v = y[i] > 0.f ? d[i] : 0.f;
    if (mask) v *= mask[i];
    d[i] = v;
  }
}
// favg[b][d] = mean_i feat[b][i][d]   (Lambda K.mean(axis=1), M:1343-1344)
__global__ __launch_bounds__(256) void tr_mean_rows_kernel(const float* __restrict__ feat, float* __restrict__ favg, int L, int D) {
  const int b = blockIdx.x;
  for (int d = threadIdx.x; d < D; d += 256) {
    float s = 0.f;
    for (int i = 0; i < L; ++i) s += feat[((size_t)b * L + i) * D + d];
    favg[(size_t)b * D + d] = s / (float)L;
  }
}
// dfeat[b][i][d] += dfavg[b][d] / L
__global__ __launch_bounds__(256) void tr_mean_rows_bwd_kernel(float* __restrict__ dfeat, const float* __restrict__ dfavg, int L, int D,
                                                               size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t b = i / ((size_t)L * D);
    dfeat[i] += dfavg[b * D + i % D] / (float)L;
  }
}
// X[t][b] = [embedding[cap_in[b][t]] | glob[b]] (adaptive, M:581 input_x: emb_off 0, glob_off E) or [glob | embedding]
// (grid-TD, the non-recurrent part of M:792: glob_off 0, emb_off E)
__global__ __launch_bounds__(256) void tr_build_x_kernel(const float* __restrict__ emb, const float* __restrict__ glob,
                                                         const int* __restrict__ cap_in, float* __restrict__ X, int B, int T, int E,
                                                         int emb_off, int glob_off) {
  const int t = blockIdx.x / B, b = blockIdx.x % B;
  const int row = cap_in[b * T + t];
  float* x = X + (size_t)blockIdx.x * 2 * E;
  for (int e = threadIdx.x; e < E; e += 256) {
    x[emb_off + e] = emb[(size_t)row * E + e];
    x[glob_off + e] = glob[(size_t)b * E + e];
  }
}
// grid-TD output of a step and its backward head: OUTm = (h2 + c_hat) * mask (M:816 + Dropout M:1299);
// DCH = dOUTm * mask, dH2tot = dH2 + DCH
__global__ __launch_bounds__(256) void tr_out_fwd_kernel(const float* __restrict__ h2, const float* __restrict__ chat,
                                                         const float* __restrict__ mask_out, float* __restrict__ OUTm, int B, int H, int T,
                                                         int t) {
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < B * H; idx += gridDim.x * 256) {
    float o = h2[idx] + chat[idx];
    if (mask_out) o *= mask_out[((size_t)(idx / H) * T + t) * H + idx % H];
    OUTm[idx] = o;
  }
}
__global__ __launch_bounds__(256) void tr_out_bwd_kernel(const float* __restrict__ dOUTm, const float* __restrict__ mask_out,
                                                         const float* __restrict__ dH2, float* __restrict__ DCH, float* __restrict__ dH2tot,
                                                         int B, int H, int T, int t) {
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < B * H; idx += gridDim.x * 256) {
    float d = dOUTm[idx];
    if (mask_out) d *= mask_out[((size_t)(idx / H) * T + t) * H + idx % H];
    DCH[idx] = d;
    dH2tot[idx] = (dH2 ? dH2[idx] : 0.f) + d;
  }
}
// LSTM-cell dropout (keras LSTMCell.call, implementation 1: one mask per gate on the inputs and on h; the wrapper calls
// the cell inside the K.rnn loop, M:582, so masks are per step): out[g][r][:] = x[r][:] * mask[t(r)][g][b(r)][:] for the
// rows r = (t, b) of `x` (rows x W); mask layout (T, 4, B, W) with t0 = first step of x.
__global__ __launch_bounds__(256) void tr_gate_masks_kernel(const float* __restrict__ x, const float* __restrict__ mask,
                                                            float* __restrict__ out, int rows, int B, int W, int t0, size_t gate_stride) {
  const size_t n = (size_t)rows * W;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t r = i / W;
    const int w = (int)(i % W), t = t0 + (int)(r / B), b = (int)(r % B);
    const float v = x[i];
#pragma unroll
    for (int g = 0; g < 4; ++g) out[g * gate_stride + i] = v * mask[(((size_t)t * 4 + g) * B + b) * W + w];
  }
}
// d[r][:] (+)= sum_g part[g][r][:] * mask[t(r)][g][b(r)][:]   (back through the four per-gate masks)
__global__ __launch_bounds__(256) void tr_gate_masks_bwd_kernel(const float* __restrict__ part, const float* __restrict__ mask,
                                                                float* __restrict__ d, int rows, int B, int W, int t0,
                                                                size_t gate_stride) {
  const size_t n = (size_t)rows * W;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t r = i / W;
    const int w = (int)(i % W), t = t0 + (int)(r / B), b = (int)(r % B);
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) s += part[g * gate_stride + i] * mask[(((size_t)t * 4 + g) * B + b) * W + w];
    d[i] = s;
  }
}

// the same for an input that is the concatenation [a | b] of two (B, H) rows (grid-TD language LSTM, x2 = [c_hat | h1]):
// out[g][b][2H]; mask (T, 4, B, 2H).  Backward: da += first halves, db += second halves of sum_g part[g] * mask.
__global__ __launch_bounds__(256) void tr_gate_masks2_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                             const float* __restrict__ mask, float* __restrict__ out, int B, int H, int t,
                                                             size_t gate_stride) {
  const int W2 = 2 * H;
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < B * W2; idx += gridDim.x * 256) {
    const int bb = idx / W2, w = idx % W2;
    const float v = w < H ? a[(size_t)bb * H + w] : b[(size_t)bb * H + w - H];
#pragma unroll
    for (int g = 0; g < 4; ++g) out[g * gate_stride + idx] = v * mask[(((size_t)t * 4 + g) * B + bb) * W2 + w];
  }
}
__global__ __launch_bounds__(256) void tr_gate_masks2_bwd_kernel(const float* __restrict__ part, const float* __restrict__ mask,
                                                                 float* __restrict__ da, float* __restrict__ db, int B, int H, int t) {
  const int W2 = 2 * H;
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < B * W2; idx += gridDim.x * 256) {
    const int bb = idx / W2, w = idx % W2;
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) s += part[(size_t)g * B * W2 + idx] * mask[(((size_t)t * 4 + g) * B + bb) * W2 + w];
    if (w < H) da[(size_t)bb * H + w] += s;
    else db[(size_t)bb * H + w - H] += s;
  }
}

// LSTM cell + sentinel of one step (keras LSTMCell.call, gate order i f c o; M:582-584).  Z row = [z_i z_f z_g z_o | u]
// before the bias; G row = activated gates, C / Hs / TC (= tanh c) / SU (= sigmoid u) / S rows of this step.
__global__ __launch_bounds__(256) void tr_cell_fwd_kernel(const float* __restrict__ Z, const float* __restrict__ bias,
                                                          const float* __restrict__ Cprev, float* __restrict__ G, float* __restrict__ C,
                                                          float* __restrict__ Hs, float* __restrict__ TC, float* __restrict__ SU,
                                                          float* __restrict__ S, int B, int H, int ldz) {
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < B * H; idx += gridDim.x * 256) {
    const int b = idx / H, h = idx % H;
    const float* z = Z + (size_t)b * ldz;
    const float i = tr_sigmoid(z[h] + bias[h]), f = tr_sigmoid(z[H + h] + bias[H + h]);
    const float g = tanhf(z[2 * H + h] + bias[2 * H + h]), o = tr_sigmoid(z[3 * H + h] + bias[3 * H + h]);
    const float c = f * (Cprev ? Cprev[idx] : 0.f) + i * g;
    const float tc = tanhf(c);
    float* gg = G + (size_t)b * 4 * H;
    gg[h] = i; gg[H + h] = f; gg[2 * H + h] = g; gg[3 * H + h] = o;
    C[idx] = c; Hs[idx] = o * tc; TC[idx] = tc;
    if (S) {                                       // cells with a visual sentinel (ldz = 5H)
      const float su = tr_sigmoid(z[4 * H + h]);
      SU[idx] = su; S[idx] = tc * su;
    }
  }
}
// Adaptive attention of one step (M:586-599), two launches:
//   scores: one wave per (caption, position): e_i = v . tanh(proj_i + h Wg)                         grid (B, ceil(L/4))
//   mix   : one workgroup per (caption, 64 hidden units): soft-max over the positions, sentinel gate beta, context,
//           OUTm = (h + c_hat) * mask_out.  The L-sized soft-max and the H-sized sentinel score are recomputed by every
//           workgroup of a caption (cheap) so that no third launch is needed.                       grid (B, H/64)
__global__ __launch_bounds__(256) void tr_att_scores_kernel(const float* __restrict__ proj, const float* __restrict__ HW,
                                                            const float* __restrict__ v, float* __restrict__ Esc, int L, int H) {
  const int b = blockIdx.x, i = blockIdx.y * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= L) return;
  const float* p = proj + ((size_t)b * L + i) * H;
  const float* hw = HW + (size_t)b * H;
  float s = 0.f;
  for (int h = lane; h < H; h += 64) s += tanhf(p[h] + hw[h]) * v[h];
  s = tr_wave_sum(s);
  if (lane == 0) Esc[(size_t)b * L + i] = s;
}
__global__ __launch_bounds__(256) void tr_att_mix_kernel(const float* __restrict__ Esc, const float* __restrict__ Vf,
                                                         const float* __restrict__ HW, const float* __restrict__ SW,
                                                         const float* __restrict__ v, const float* __restrict__ Hs,
                                                         const float* __restrict__ S, const float* __restrict__ mask_out,
                                                         float* __restrict__ ALPHA, float* __restrict__ BETA, float* __restrict__ CTX,
                                                         float* __restrict__ OUTm, int L, int H, int T, int t) {
  extern __shared__ float sm[];                 // e[L] | red[8] | part[256]
  float* e = sm;
  float* red = sm + L;
  float* part = sm + L + 8;
  const int b = blockIdx.x, h0 = blockIdx.y * 64, tid = threadIdx.x;
  const float* hw = HW + (size_t)b * H;
  float zs = 0.f;
  for (int h = tid; h < H; h += 256) zs += tanhf(SW[(size_t)b * H + h] + hw[h]) * v[h];
  zs = tr_block_sum(zs, red);
  float mx = -INFINITY;
  for (int i = tid; i < L; i += 256) { e[i] = Esc[(size_t)b * L + i]; mx = fmaxf(mx, e[i]); }
  mx = tr_block_max(mx, red);
  float se = 0.f;
  for (int i = tid; i < L; i += 256) se += expf(e[i] - mx);
  se = tr_block_sum(se, red);
  const float mx2 = fmaxf(mx, zs);
  const float beta = expf(zs - mx2) / (se * expf(mx - mx2) + expf(zs - mx2));
  for (int i = tid; i < L; i += 256) {
    const float a = expf(e[i] - mx) / se;
    e[i] = a;
    if (blockIdx.y == 0) ALPHA[(size_t)b * L + i] = a;
  }
  if (blockIdx.y == 0 && tid == 0) BETA[b] = beta;
  __syncthreads();
  const int hl = tid & 63, q = tid >> 6, h = h0 + hl;
  float c = 0.f;
  if (h < H)
    for (int i = q; i < L; i += 4) c += e[i] * Vf[((size_t)b * L + i) * H + h];
  part[tid] = c;
  __syncthreads();
  if (q == 0 && h < H) {
    c = part[hl] + part[64 + hl] + part[128 + hl] + part[192 + hl];
    const size_t o = (size_t)b * H + h;
    CTX[o] = c;
    float out = (Hs ? Hs[o] : 0.f) + beta * S[o] + (1.f - beta) * c;       // Hs null: c_hat alone (grid-TD)
    if (mask_out) out *= mask_out[((size_t)b * T + t) * H + h];
    OUTm[o] = out;
  }
}
// Two-headed loss (M:95-103, :1364-1373) of one (t, b) row, in place: Z row (logits before the bias) -> d loss / d logits.
// part[row] = (CE head 1, CE head 2, hit head 1, hit head 2, labelled) — hit = the label is the arg-max (M:105-124).
// Rows of the last time step and rows without a label give zero.
// ml: Dropout on the logits (grid-TD model, M:1303-1304), (B, T, V) or null.
__global__ __launch_bounds__(256) void tr_loss_kernel(float* __restrict__ Z, const float* __restrict__ bout, const float* __restrict__ lw,
                                                      const float* __restrict__ ml, const int* __restrict__ y_idx,
                                                      float* __restrict__ part, int B, int T, int V, float scale) {
  __shared__ float red[4];
  const int row = blockIdx.x, t = row / B, b = row % B, tid = threadIdx.x;
  float* z = Z + (size_t)row * V;
  const int y = y_idx[b * T + t];
  if (t == T - 1 || y < 0) {
    for (int k = tid; k < V; k += 256) z[k] = 0.f;
    if (tid < 5) part[5 * row + tid] = 0.f;
    return;
  }
  const float* w = lw + ((size_t)b * T + t) * V;
  const float* mk = ml ? ml + ((size_t)b * T + t) * V : nullptr;
  auto logit = [&](int k) { const float x = z[k] + bout[k]; return mk ? x * mk[k] : x; };
  float m1 = -INFINITY, m2 = -INFINITY;
  for (int k = tid; k < V; k += 256) {
    const float x = logit(k);
    m1 = fmaxf(m1, x); m2 = fmaxf(m2, x * w[k]);
  }
  m1 = tr_block_max(m1, red);
  m2 = tr_block_max(m2, red);
  float s1 = 0.f, s2 = 0.f;
  for (int k = tid; k < V; k += 256) {
    const float x = logit(k);
    s1 += expf(x - m1); s2 += expf(x * w[k] - m2);
  }
  s1 = tr_block_sum(s1, red);
  s2 = tr_block_sum(s2, red);
  const float zy = logit(y);
  __syncthreads();
  if (tid == 0) {
    part[5 * row] = -(zy - m1 - logf(s1));
    part[5 * row + 1] = -(zy * w[y] - m2 - logf(s2));
    part[5 * row + 2] = zy >= m1 ? 1.f : 0.f;
    part[5 * row + 3] = zy * w[y] >= m2 ? 1.f : 0.f;
    part[5 * row + 4] = 1.f;
  }
  for (int k = tid; k < V; k += 256) {
    const float x = logit(k);
    const float d = k == y ? 1.f : 0.f;
    const float g = scale * (0.5f * (expf(x - m1) / s1 - d) + 0.5f * w[k] * (expf(x * w[k] - m2) / s2 - d));
    z[k] = mk ? g * mk[k] : g;
  }
}
// losses[0..4] = (0.5 l1 + 0.5 l2, l1, l2, accuracy head 1, accuracy head 2) — what train_on_batch returns
// (train.py:578-579); rows summed in index order
__global__ void tr_loss_final_kernel(const float* __restrict__ part, int rows, float scale, float* __restrict__ losses) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double a = 0.0, c = 0.0, h1 = 0.0, h2 = 0.0, n = 0.0;
  for (int r = 0; r < rows; ++r) {
    a += part[5 * r]; c += part[5 * r + 1]; h1 += part[5 * r + 2]; h2 += part[5 * r + 3]; n += part[5 * r + 4];
  }
  losses[1] = (float)(a * scale); losses[2] = (float)(c * scale);
  losses[0] = 0.5f * losses[1] + 0.5f * losses[2];
  losses[3] = n > 0.0 ? (float)(h1 / n) : 0.f;
  losses[4] = n > 0.0 ? (float)(h2 / n) : 0.f;
}

// Backward of the attention step, three launches.  In: dOUTm row of the step, carry dH.
//   head (grid B): dHtot = dH + dout, dS = beta dout, dCtx = (1 - beta) dout, dBeta = dout . (s - ctx)
//   dalpha (grid (B, ceil(L/4))): DA_i = dCtx . Vf_i, one wave per position
//   main (grid (B, H/64)): scores' gradient de_i (soft-max over L and the sentinel soft-max, recomputed per workgroup),
//        then for its 64 hidden units: dProj += de_i v (1 - A^2), dVf += alpha_i dCtx, DHW (d of h Wg), DZS (d of the
//        sentinel score's pre-tanh row), dVacc (the attention vector's gradient, per caption).
__global__ __launch_bounds__(256) void tr_att_bwd_head_kernel(const float* __restrict__ S, const float* __restrict__ CTX,
                                                              const float* __restrict__ BETA, const float* __restrict__ dOUTm,
                                                              const float* __restrict__ mask_out, const float* __restrict__ dH,
                                                              float* __restrict__ dHtot, float* __restrict__ dS, float* __restrict__ dCtx,
                                                              float* __restrict__ dBeta, int H, int T, int t, int add_to_h) {
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float beta = BETA[b];
  float dbeta = 0.f;
  for (int h = tid; h < H; h += 256) {
    const size_t o = (size_t)b * H + h;
    float d = dOUTm[o];
    if (mask_out) d *= mask_out[((size_t)b * T + t) * H + h];
    dHtot[o] = (dH ? dH[o] : 0.f) + (add_to_h ? d : 0.f);     // adaptive: out = h + c_hat; grid-TD: d is d c_hat only
    dS[o] = beta * d;
    dCtx[o] = (1.f - beta) * d;
    dbeta += d * (S[o] - CTX[o]);
  }
  dbeta = tr_block_sum(dbeta, red);
  if (tid == 0) dBeta[b] = dbeta;
}
__global__ __launch_bounds__(256) void tr_att_bwd_dalpha_kernel(const float* __restrict__ Vf, const float* __restrict__ dCtx,
                                                                float* __restrict__ DA, int L, int H) {
  const int b = blockIdx.x, i = blockIdx.y * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= L) return;
  const float* vf = Vf + ((size_t)b * L + i) * H;
  const float* dc = dCtx + (size_t)b * H;
  float s = 0.f;
  for (int h = lane; h < H; h += 64) s += dc[h] * vf[h];
  s = tr_wave_sum(s);
  if (lane == 0) DA[(size_t)b * L + i] = s;
}
__global__ __launch_bounds__(256) void tr_att_bwd_main_kernel(const float* __restrict__ proj, const float* __restrict__ HW,
                                                              const float* __restrict__ SW, const float* __restrict__ v,
                                                              const float* __restrict__ ALPHA, const float* __restrict__ BETA,
                                                              const float* __restrict__ DA, const float* __restrict__ dBeta,
                                                              const float* __restrict__ dCtx, float* __restrict__ DZS,
                                                              float* __restrict__ DHW, float* __restrict__ dProj, float* __restrict__ dVf,
                                                              float* __restrict__ dVacc, int L, int H) {
  extern __shared__ float sm[];                 // de[L] | alpha[L] | red[8] | part[512]
  float* de = sm;
  float* al = sm + L;
  float* red = sm + 2 * L;
  float* part = sm + 2 * L + 8;
  const int b = blockIdx.x, h0 = blockIdx.y * 64, tid = threadIdx.x;
  const float beta = BETA[b], dbeta = dBeta[b];
  float dot = 0.f;
  for (int i = tid; i < L; i += 256) {
    al[i] = ALPHA[(size_t)b * L + i];
    de[i] = DA[(size_t)b * L + i];
    dot += al[i] * de[i];
  }
  dot = tr_block_sum(dot, red);
  for (int i = tid; i < L; i += 256) de[i] = al[i] * (de[i] - dot) - beta * (1.f - beta) * al[i] * dbeta;
  const float dzs = beta * (1.f - beta) * dbeta;
  __syncthreads();
  const int hl = tid & 63, q = tid >> 6, h = h0 + hl;
  float dhw = 0.f, dv = 0.f;
  if (h < H) {
    const float vh = v[h], hwh = HW[(size_t)b * H + h], dc = dCtx[(size_t)b * H + h];
    for (int i = q; i < L; i += 4) {
      const size_t o = ((size_t)b * L + i) * H + h;
      const float A = tanhf(proj[o] + hwh);
      const float dA = de[i] * vh * (1.f - A * A);
      dProj[o] += dA;
      dVf[o] += al[i] * dc;
      dhw += dA;
      dv += de[i] * A;
    }
  }
  part[tid] = dhw;
  part[256 + tid] = dv;
  __syncthreads();
  if (q == 0 && h < H) {
    dhw = part[hl] + part[64 + hl] + part[128 + hl] + part[192 + hl];
    dv = part[256 + hl] + part[320 + hl] + part[384 + hl] + part[448 + hl];
    const size_t o = (size_t)b * H + h;
    const float vh = v[h];
    const float As = tanhf(SW[o] + HW[o]);
    const float dz = dzs * vh * (1.f - As * As);
    DZS[o] = dz;
    DHW[o] = dhw + dz;
    dVacc[o] += dv + dzs * As;
  }
}
// Backward of tr_cell_fwd_kernel: dHtot, dS, carry dC -> DZ row [d z_i, d z_f, d z_g, d z_o | d u], carry dC (in place).
__global__ __launch_bounds__(256) void tr_cell_bwd_kernel(const float* __restrict__ G, const float* __restrict__ Cprev,
                                                          const float* __restrict__ TC, const float* __restrict__ SU,
                                                          const float* __restrict__ dHtot, const float* __restrict__ dS,
                                                          float* __restrict__ dC, float* __restrict__ DZ, int B, int H, int ldz) {
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < B * H; idx += gridDim.x * 256) {
    const int b = idx / H, h = idx % H;
    const float* gg = G + (size_t)b * 4 * H;
    const float i = gg[h], f = gg[H + h], g = gg[2 * H + h], o = gg[3 * H + h];
    const float tc = TC[idx], su = dS ? SU[idx] : 0.f, ds = dS ? dS[idx] : 0.f, dh = dHtot[idx];
    const float dct = dC[idx] + (ds * su + dh * o) * (1.f - tc * tc);
    float* dz = DZ + (size_t)b * ldz;
    dz[h] = dct * g * i * (1.f - i);
    dz[H + h] = dct * (Cprev ? Cprev[idx] : 0.f) * f * (1.f - f);
    dz[2 * H + h] = dct * i * (1.f - g * g);
    dz[3 * H + h] = dh * tc * o * (1.f - o);
    if (dS) dz[4 * H + h] = ds * tc * su * (1.f - su);
    dC[idx] = dct * f;
  }
}
// dglob[b][e] = sum_t dX[t][b][E + e]
__global__ __launch_bounds__(256) void tr_dglob_kernel(const float* __restrict__ dX, float* __restrict__ dglob, int B, int T, int E,
                                                       int glob_off) {
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < B * E; idx += gridDim.x * 256) {
    const int b = idx / E, e = idx % E;
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += dX[((size_t)t * B + b) * 2 * E + glob_off + e];
    dglob[idx] = s;
  }
}
// Embedding gradient: dEmb[row] = sum of the dX word halves of every (t, b) that read that row, in (t, b) order
// (one workgroup per (t, b); the first reader of a row sums for all of them — no atomics, run-to-run identical).
__global__ __launch_bounds__(256) void tr_embedding_bwd_kernel(const float* __restrict__ dX, const int* __restrict__ cap_in,
                                                               float* __restrict__ dEmb, int B, int T, int E, int emb_off) {
  const int me = blockIdx.x, n = B * T;
  const int row = cap_in[(me % B) * T + me / B];
  for (int j = 0; j < me; ++j)
    if (cap_in[(j % B) * T + j / B] == row) return;
  for (int e = threadIdx.x; e < E; e += 256) {
    float s = 0.f;
    for (int j = me; j < n; ++j)
      if (cap_in[(j % B) * T + j / B] == row) s += dX[(size_t)j * 2 * E + emb_off + e];
    dEmb[(size_t)row * E + e] = s;
  }
}
// Image-layer weight gradient from the product over the im2col matrix A1 = [x+ patch(27) | 0 | x- patch(27) | 0]:
// dW[k][co] = G[k][co] + G[32 + k][co], k = tap * 3 + c < 27   (G = A1^T dZ, 64 x Cout)
__global__ __launch_bounds__(256) void tr_fold_image_wgrad_kernel(const float* __restrict__ G, float* __restrict__ dW, int Cout) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < 27 * Cout) dW[i] = G[i] + G[32 * Cout + i];
}

// keras Adam with clipvalue (optimizers.py: clip, moments, update); lr_t = lr sqrt(1 - b2^t) / (1 - b1^t) from the host
__global__ __launch_bounds__(256) void tr_adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                      float* __restrict__ v, size_t n, float lr_t, float clip, float b1, float b2,
                                                      float eps) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float gi = g[i];
    if (clip > 0.f) gi = fminf(fmaxf(gi, -clip), clip);
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    p[i] -= lr_t * mi / (sqrtf(vi) + eps);
  }
}

}  // namespace lrp
