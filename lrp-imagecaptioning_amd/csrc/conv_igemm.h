// conv_igemm.h — MFMA implicit-GEMM 3x3/1x1 convolution for gfx950 with the LRP epilogues fused.  This one kernel
// family carries >95 % of the FLOPs of the hot path:
//   * encoder forward: activation chain (EPI_BIAS / EPI_BIAS_RELU, three-way split operands, TERMS 15 + 3) and
//     denominators Z+ (EPI_BIAS, TERMS 7); exact-fp32 variants (EPI_FWD_DUAL)             — per image, cached
//   * conv-LRP alpha1beta0 backward (EPI_MUL / EPI_MUL_UP2 / EPI_IMG_STENCIL)              — per token
//     RR:274-322 restructured: S_{l-1} = up2?(convT(S_l, w_l+)) * G_{l-1}
//   * every dense product of the decoder LRP / gradient paths (taps = 1, EPI_STORE / EPI_MUL)
// Template axes: tile (WM, WN, TM, TN), epilogue EPI, operand arithmetic PREC (exact fp32 MFMA | split-bf16),
// HALO (3x3: A tile + halo resident in LDS for all 9 taps), BREG (N <= 64: weights in registers, no barrier per tap),
// TERMS (which partial products of the split operands are issued).  DESIGN.md 4.1 has the measurements behind each.
//
// GEMM view: D[m][n] = sum_k A[m][k] * B[k][n],  m = output pixel (NHWC row),
// n = output channel, k = (tap, input channel).  A is gathered on the fly
// (im2col never materialised); B is pre-packed [n][k] so both operands are
// "rows of 32 consecutive k" = 128 B, staged through LDS unpadded with an XOR
// swizzle of the 16 B chunk index, chunk ^ ((row>>1)&7): conflict-free for the
// ds_read_b128 fragment reads AND the ds_write_b128 staging writes.
//
// MFMA: v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD).  Lane l feeds
// A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31]; we let lane-half h consume
// k = 4h+s at sub-step s so that one ds_read_b128 per operand covers 4 MFMAs
// (any k permutation is legal as long as A and B agree).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

#include "common.h"

namespace lrp {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// PREC_FP32 : operands fp32, v_mfma_f32_32x32x2_f32 (exact fp32 fma chain).  The walk and the forward of LRP_PREC_FP32; the image
//   layer's forward GEMM and the decoder's products in every mode.
// PREC_BF16X3: operands stored as "split8" — per 8 consecutive channels 32 B = [8 x bf16 hi | 8 x bf16 lo] with
//   x ~= hi + lo (16 mantissa bits), same bytes as fp32.  Each product is hi*hi' + hi*lo' + lo*hi' on
//   v_mfma_f32_32x32x16_bf16 with fp32 accumulation: 3 MFMAs of 32 cycles per 16 k instead of 8 of 64
//   (5.3x fewer matrix-pipe cycles).  The per-token reverse walk of the DEFAULT mode (lrp_create: LRP_PREC_BF16X3): worst case
//   2^-16 per product whatever the weights [MI355X: 2-3.5e-6 relative L1 vs the float64 graph on He-normal kernels,
//   6e-6 ... 1.6e-5 on trained-like ones, DESIGN 4.2].
// PREC_F16X2 : both operands as fp16 pairs hi + lo (22 mantissa bits) in the same split8 layout, carried scaled by powers
//   of two taken from MEASURED maxima (fp16 has 5 exponent bits): per token for the relevance S of the reverse walk
//   (ConvArgs::tok_*; Encoder::explain), per image for the forward's activations, per matrix for the weights.
//   TERMS 7: hi*hi' + hi*lo' + lo*hi', three v_mfma_f32_32x32x16_f16 per 16 k — fp32-grade.  This is the encoder FORWARD of every
//   mode but LRP_PREC_FP32 (the interleaved dual conv a_l | Z+_l, blocked accumulation: features 7e-7 from float64) and the top
//   block of the opt-in fast walk.
//   TERMS 5: the weights' lo half is not read: lo*w + hi*w, TWO MFMAs — the layers of the OPT-IN fast walk (LRP_PREC_F16X2) that
//   lrp_set_fast_layers / calibration.py selected; TERMS bit 4 computes the matching two-term denominators Z+ in the forward.
//   One fp16 per weight is 2^-12 per product: fine on dense Gaussian kernels (the rounding averages out), above the 1e-4 bar on
//   sparse heavy-tailed ones — which is why this is not the default (tests/test_gpu_stress_parity.py).
enum ConvPrec { PREC_FP32 = 0, PREC_BF16X3 = 1, PREC_F16X2 = 2 };

__device__ __forceinline__ void split8h_store(const float* r, float* dst) {  // 8 fp32 -> 32 B [fp16 hi8 | fp16 lo8]
  f16x8 hi, lo;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    hi[q] = (_Float16)r[q];
    lo[q] = (_Float16)(r[q] - (float)hi[q]);
  }
  u32x4* d = reinterpret_cast<u32x4*>(dst);
  d[0] = __builtin_bit_cast(u32x4, hi);
  d[1] = __builtin_bit_cast(u32x4, lo);
}
__device__ __forceinline__ void split8_store(const float* r, float* dst) {   // 8 fp32 -> 32 B [hi8 | lo8]
  bf16x8 hi, lo;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    hi[q] = (__bf16)r[q];
    lo[q] = (__bf16)(r[q] - (float)hi[q]);
  }
  u32x4* d = reinterpret_cast<u32x4*>(dst);
  d[0] = __builtin_bit_cast(u32x4, hi);
  d[1] = __builtin_bit_cast(u32x4, lo);
}

enum ConvEpi {
  EPI_BIAS_RELU = 0,  // out = relu(acc + bias)
  EPI_BIAS = 1,       // out = acc + bias
  EPI_MUL = 2,        // out = acc * aux[img(row)]               (conv-LRP, no pool)
  EPI_MUL_UP2 = 3,    // out(2x res) = acc * aux[img(row)](2x)   (conv-LRP through a 2x2 max-pool)
  EPI_FWD_DUAL = 4,   // cols [0,split): out = relu(acc+bias); cols [split,2split): out2 = acc+bias  (a_l and Z+_l)
  EPI_STORE = 5,      // out = acc (row stride = N)   (image layer: tap-expanded channel reduction, see img_stencil_kernel)
  // Image layer in ONE launch: the rows of a tile are a 16 x 16 pixel PATCH (14 x 14 output pixels + 1 halo), the
  // tile computes T = S_1 . W (54 tap-expanded columns, see cnn_kernels.h) for the patch, keeps it in LDS and applies
  // the 9-tap shift-and-add for its 14 x 14 interior: S_1 is read once (x 1.31 halo) and only R_img is written,
  // instead of writing and re-reading the 3.5 GB T tensor.  BM = 256, BN = 64, 1 tap.
  EPI_IMG_STENCIL = 6
};
constexpr int IMG_PATCH = 16, IMG_TILE = 14;

// Halo-resident launches walk the image STACK (all tokens on top of each other) tile row by tile row.  The gate a tile is
// multiplied with belongs to the token's IMAGE, and with several tokens per image (10 words per caption) the same gate rows
// were fetched once per token — the stack order puts 25 tile rows of other traffic between two uses, far more than an
// XCD's 4 MB of L2 [MI355X, block1_conv2: FETCH_SIZE 8.2 GB = S_in 4.1 + 10 x 0.41 of gates].  TileOrder is a per-layer
// host cache of a permutation of the tiles — ordered by (image, column strip, row band, token) — so that the tiles of an
// image's tokens at the same place run back to back on one XCD (they share the gate rows in its L2) and a strip's next
// row band follows within a few dozen tiles (the vertical halo rows of S are still there); built from the call's
// token -> image map, rebuilt only when that map changes.  Tiles are independent, so the order changes no result bit.
struct TileOrder {
  std::vector<int> sig;                                 // token -> image map the table was built for
  int H = 0, th = 0, nyh = 0, cols_t = 0, tpt = 0;
  bool identity = true;
  int* dev = nullptr;
  int* pinned = nullptr;
  size_t cap = 0;
  hipEvent_t ev = nullptr;                              // the last upload
  TileOrder() = default;
  TileOrder(const TileOrder&) = delete;
  TileOrder& operator=(const TileOrder&) = delete;
  ~TileOrder() {
    if (dev) (void)hipFree(dev);
    if (pinned) (void)hipHostFree(pinned);
    if (ev) (void)hipEventDestroy(ev);
  }
};

struct ConvArgs {
  const float* in;     // [NB][H][W][Cin] fp32
  const float* wpk;    // [n_tiles*BN][K] fp32, K = taps*CinP, CinP = roundup(Cin,32), zero padded
  const float* wpk_frag;  // BREG: the same weights fragment-major, [kc][step][hi|lo][lane half][BN = 64] x 16 B
  int NB, H, W, Cin, CinP;
  int N;               // valid output columns
  int taps;            // 9 (3x3 same) or 1
  int M;               // NB*H*W
  int m_tiles, n_tiles;
  const float* bias;
  float* out;
  float* out2;
  const float* aux;
  const int* row2img;  // per input image-slot n -> cache slot (nullptr = identity)
  int split;
  int out_plain;       // bf16x3 MUL epilogues: 1 = write fp32 instead of re-splitting (last GEMM of a chain)
  // EPI_IMG_STENCIL: tiles per image row / column, ximg = the images (x of the image layer), mode 0 LRP | 1 sum | 2 x*sum
  int tiles_x, tiles_y, img_mode;
  const float* ximg;
  const float* addend; // BIAS / BIAS_RELU epilogues: out = [relu](acc + bias + addend)  (bias may be null)
  // EPI_BIAS only: out = gate_src / SafeDivide-denominator(acc + bias) — the relevance gate G_l = a_l / safe(Z+_l)
  // (IL:456-458) straight from the denominator conv, Z+_l itself never goes to memory.  gate_src may alias out (the
  // overlapped encode parks a_l in the gate's storage): an element is read and written by the same thread.
  const float* gate_src;
  // EPI_MUL only — tail of a residual block in one epilogue (ResNet walk):
  //   r    = acc * aux[img] + join[row] * join_gate[img]     (the shortcut's share joins the main branch, KG:799-803)
  //   out  = r                                               (fp32 / split8 as usual)
  //   out2 = r * gate2[img]                                  (head of the NEXT block's conv chain; split8 in bf16x3 mode)
  const float* join;
  const float* join_gate;
  const float* gate2;
  float* out2s;
  // gradient baselines on the MUL epilogues: the cached LRP gate is used as a MASK (gate != 0 <=> the unit's ReLU was
  // active and it won its pool window), and guided backprop also clamps the propagated value at 0
  int gate_binary, relu_out;
  int dual_norelu;     // EPI_FWD_DUAL: first half without the relu (a conv + BatchNorm unit: c and Z+ in one pass)
  // EPI_FWD_DUAL, interleaved weights (dual_il): the stacked matrix has its rows in blocks of 32 — [w of channels 32g..32g+31 |
  // w+ of the SAME channels] — so a tile holds c and Z+ of a channel side by side and the epilogue can finish the pair:
  // dual_gate = 1: out = a_l = relu(c), out2 = the relevance gate G_l = a_l / safe(Z+_l) (IL:456-458); Z+_l never goes to
  // memory and no gate pass follows.  dual_gate = 0: out2 = Z+_l as in the stacked form.  Needs split % 32 == 0.
  int dual_il, dual_gate;
  // dual_gate = 2 (a conv + BatchNorm unit of the ResNet encoder, RA:197-257 folded with alpha1beta0; resnet_kernels.h
  // rn_bn_unit_kernel is the unfused form): with c = conv + b, Z = the alpha1beta0 denominator,
  //   y = gamma (c - mean) / sqrt(var + eps) + beta,  Q = c (y - beta) / stab((c - mean) y) / safe(Z)
  //   bn_relu: out = relu(y), out2 = relu(y) Q      else: out = y, out2 = Q
  const float* bn_gamma; const float* bn_beta; const float* bn_mean; const float* bn_var;
  float bn_eps; int bn_relu;
  // halo-resident 3x3 variant (template HALO): a tile is th rows x tw (<= 14) columns of the image stack
  // (all NB images on top of each other: Y = n*H + h), cols_t tiles per image row; hrows = rows of the
  // resident image (th + 2 + separator rows), each HALO_PITCH pixels wide
  int tw, th, hrows, cols_t, nyh;
  const int* tile_map;        // device: launch position -> tile of the stack (nullptr = stack order); filled by the launcher from:
  TileOrder* order;           // host: the layer's cache (nullptr = never reorder)
  const int* row2img_host;    // host copy of row2img for this call (nullptr = identity: one token per image)
  // PREC_F16X2: per-token power-of-two scaling of the fp16 relevance tensors (indexed by the token slot n of a row).
  //   stored input = true * 2^e_in[n], |stored input| <= max_in[n] (measured by the producer).  tok_scale_kernel
  //   (cnn_kernels.h) picks k[n] = floor(log2(30000 / (max_in[n] * wnorm))) — wnorm = max row sum of |w| of the layer's
  //   backward matrix, the gate is <= 1, so |acc * gate| * 2^k stays below fp16's 65504 — and hands this launch
  //   tok_fac[n] = 2^k[n]; the epilogue stores acc * gate * tok_fac[n] and raises tok_max_out[n] (float bits) to the
  //   largest |stored output|.  EPI_IMG_STENCIL ends the chain: tok_fac[n] = 2^-e_in[n].
  const float* tok_fac;
  unsigned* tok_max_out;
  // PREC_F16X2 forward (EPI_BIAS / EPI_BIAS_RELU, TERMS 7 = fp16 pairs on BOTH operands, 22 mantissa bits: an fp32-grade
  // product in three MFMAs): the input tensor is stored scaled by a power of two; *in_unscale (device scalar) = its
  // inverse, applied to the accumulator in front of the bias.  act_max_out: ACT_MAX_SLOTS float-bit slots raised to the
  // largest |out| (slot = block id mod slots: spreads the atomics; the consumer takes the maximum over the slots).
  const float* in_unscale;
  unsigned* act_max_out;
  // EPI_FWD_DUAL, interleaved rows, PREC_F16X2: the epilogue also writes a_l as the NEXT conv's operand — fp16 pairs
  // [hi8 | lo8] of a_l * *pairs_scale (a power of two chosen BEFORE the launch from a bound on |a_l|: the measured maximum
  // of this conv's input x its weights' largest absolute row sum + max|b|; cnn_kernels.h fwd_scale_kernel) — so no split
  // pass runs between two convs.  skip_out: a_l itself (fp32) is not written (nobody else reads it).
  float* pairs_out;
  const float* pairs_scale;
  int skip_out;
  // scale_per_img (interleaved dual forward): in_unscale, pairs_scale and act_max_out are arrays indexed by the IMAGE of a row
  // (row / img_rows; act_max_out[image][ACT_MAX_SLOTS]) instead of one record for the call — an image's rows only ever gather
  // from that image, so a scale per image is as legal as one per call, and the result for an image no longer depends on
  // which other images share its batch.
  int scale_per_img, img_rows, n_imgs;
  // Compact pool interface (PREC_BF16X3; first form: the weights-in-registers kernel, Cin <= 64, fields up2_src / up2_gate /
  // up2_gc; general form: every halo kernel through up2_pairs below): the relevance entering this layer came
  // through a 2x2 max-pool, i.e. S_in[n][y][x][c] = P[n][y/2][x/2][c] * G_up[img(n)][y][x][c] with exactly one non-zero per
  // window and channel.  Instead of reading that 4x-expanded, 75 %-zero tensor (which its producer would have had to write),
  // the tile's resident image is BUILT from P (fp32, pooled resolution, written by the producer with gate_none) and the
  // pool gate of this layer's output (fp32 per image, shared by an image's tokens in L2): `in` is then unused.
  const float* up2_src;        // P  [NB][H/2][W/2][Cin] fp32
  const float* up2_gate;       // G_up [images][H][W][Cin] fp32
  int gate_none;               // EPI_MUL: out = acc (no gate: the consumer applies it, see up2_src)
  // The pool gate in compact form as well (per-token tiles, ConvArgs::tpt > 0): gc [images][H/2][W/2][Cin] = the window's one
  // non-zero gate value, gpos (same shape, bytes) = its position 2 dy + dx.  The prologue then walks WINDOWS instead of pixels:
  // P, gc and gpos once per window and channel group (a quarter of the loads of the full-resolution gate, which every pixel
  // item fetched together with the same P again) and writes the product to the one position that has it, zeros to the others.
  const float* up2_gc;
  const unsigned char* up2_gpos;
  // up2_pairs = 1 (pipelined halo kernels): up2_src holds S_c = acc x compact gate at pooled resolution as bf16 pairs (its
  // producer ran EPI_MUL with the COMPACT gate as its gate): the consumer needs only the position bytes
  int up2_pairs;
  // Image layer folded into the epilogue of the layer above it (weights-in-registers kernel, PREC_BF16X3, N = 64, EPI_MUL):
  // S_1 = acc x gate never goes to memory.  The tile turns it into bf16 pairs in LDS, multiplies it with the tap-expanded
  // 64 -> 54 matrix `img_w` (the image layer's T = S_1 . W, cnn_kernels.h) and applies the 9-tap shift-and-add for the
  // SOURCE pixels it owns: per tile (th + 2) x (tw + 2) output positions (its pixels and the one-pixel ring around them)
  // x 6 partial sums go to img_part[tile of the stack][position][6]; img_partial_sum_kernel adds the up to four tiles'
  // partials of a pixel in a fixed order and applies x+ / x-.  Tiles are laid out per token (tpt below: none straddles two
  // tokens, so the grouping of a pixel's nine taps into partials is the same for every token and every batch: results stay
  // batch-invariant bit for bit).
  const float* img_w;          // [64][64] split8 bf16 pairs (layer 0's backward matrix)
  float* img_part;
  // tpt > 0 (folded launch): tiles are laid out PER TOKEN — tpt tile rows of th stack rows per token, the last one possibly
  // short — instead of over the whole stack, so that no tile straddles two tokens and the tile boundaries fall at the same
  // image rows for every token.  Tile row R of the launch covers token R / tpt, image rows (R % tpt) * th ...
  int tpt;
  int epi_generic;             // MUL / MUL_UP2 epilogues: 1 = always the general pass loop (A/B switch LRP_EPI_FAST=0; filled by the launcher)
  // 2x2 max-pool fused into the interleaved dual forward's epilogue (PREC_F16X2, 128 x 128 resident-image tile: the whole C
  // tile is in LDS; tiles of an EVEN number of rows and columns starting at even positions, so a window never straddles two
  // tiles): per window and channel the epilogue takes the first maximum in scan order, and writes — instead of a_l and Z+_l
  // at full resolution, which a streaming pass then read back (cnn_kernels.h pool_gate_split_kernel: the same arithmetic) —
  //   pool_gc   [images][H/2][W/2][C]   the gate a_l / safe(Z+_l) at the winning position (IL:456-458 through the pool)
  //   pool_pos  same shape, bytes        that position, 2 dy + dx
  //   pairs_out [images][H/2][W/2][C]   the pooled activation as the next conv's fp16 pairs (x pairs_scale[image])
  //   pool_x    (optional) the pooled activation in fp32
  // and raises act_max_out from the pooled values.  The full-resolution gate is rebuilt from (pool_gc, pool_pos) on demand.
  float* pool_gc;
  unsigned char* pool_pos;
  float* pool_x;
};
constexpr int ACT_MAX_SLOTS = 64;

constexpr int LDS_STRIDE = 32;   // floats per staged row (128 B, no padding; swizzled chunks)

// Out-of-image taps / ragged tails read this instead of branching around the load: the
// select is two v_cndmask on the address, the load itself stays unconditional and the
// eight loads of a chunk issue back to back.
__device__ __attribute__((aligned(16))) float lrp_zero_page[4] = {0.f, 0.f, 0.f, 0.f};   // non-const: stays in the GLOBAL address space (a const page makes the select generic -> flat_load, which also counts on lgkmcnt)

// waves per SIMD the register allocator must leave room for: LDS already limits a CU to
// floor(160 KB / LDS per block) blocks of NW waves
constexpr int conv_min_waves(int NW, int TM, int TN, bool halo = false) { return NW == 8 || halo ? 2 : (TM * TN >= 4 ? 2 : 3); }

// HALO (3x3 only): instead of re-staging the A tile for each of the 9 taps (9 x BM rows per 32-channel
// chunk, 8 of them L2 hits but still L2->LDS traffic, the limiter of the bf16x3 mode), the tile's pixels
// PLUS their one-pixel halo are staged once per channel chunk and the taps read shifted rows of that
// resident image.  A traffic per chunk: 9 x 256 rows -> 352 rows.
// The resident image is a window of the EXTENDED stack: every image contributes its H rows plus one all-zero
// separator row (extended row E = n*(H+1) + h, h == H is the separator), and columns x0-1 .. x0+tw with zeros
// outside [0, W).  A pixel's 3x3 neighbourhood is then always at fixed offsets (dy*HALO_PITCH + dx) from it,
// borders included: no per-lane tap masks in the main loop.  See DESIGN.md 4.1.
constexpr int HALO_PITCH = 16, HALO_PL = 4;
constexpr int conv_halo_rows(int BM) { return BM == 256 ? 352 : 192; }   // x 128 B; 22 / 12 image rows

// BREG (HALO, bf16x3, N <= 64 layers): the weights never touch LDS.  With K <= 2 x 32 channels per group the whole A
// operand of a group is resident (both LDS buffers hold one 32-channel chunk each), and the B fragments of a
// (tap, chunk) are 4 coalesced 16 B loads per lane from a fragment-major copy of the packed weights (wpk_frag,
// L1/L2 resident: 8 KB per tap), prefetched one tap ahead in registers.  The main loop then has NO barrier per tap —
// only one per channel group — which is what the 12-MFMA-per-tap waves of the N = 64 tiles could not amortise.
// TERMS (bf16x3 operand format only): which of the four partial products of (ah + al)(bh + bl) are issued —
// bit 0: al*bh, bit 1: ah*bl, bit 2: ah*bh, bit 3: al*bl.  7 = the split-bf16 product of the reverse walk.  The
// exact forward uses two passes over THREE-way split operands x = h + m + l (24 mantissa bits):
//   pass A  (h|m) x (h|m), TERMS 15: hh + hm + mh + mm      pass B  (h|l) x (h|l), TERMS 3: hl + lh
// = every partial product down to 2^-16 of the leading one, i.e. an fp32-grade product in 6 bf16 MFMAs of 32 cycles
// per 16 k (192) instead of 8 fp32 MFMAs of 64 (512).
// NS (plain staging only: !HALO, !BREG): LDS stages of the k pipeline.  2 = chunk kc+2 is launched at the barrier of iteration kc
// and must have landed one iteration later — fine when an iteration holds 24-48 MFMAs per wave, but the 64 x 64 tiles of the
// small-grid launches (6 MFMAs per wave and iteration) then run at one L2 / HBM round trip per k-step [MI355X, one image:
// 0.65 us per k-step, 95 us for a K = 4608 launch].  With NS stages NS-1 chunks are in flight and the barrier waits with a
// COUNTED s_waitcnt vmcnt((NS-2) x DMA instructions per chunk) for the oldest only.
template <int WM, int WN, int TM, int TN, int EPI, int PREC, bool HALO = false, bool BREG = false, int TERMS = 7, int NS = 2>
__global__ __launch_bounds__(64 * WM * WN, BREG ? 3 : NS > 2 ? 2 : conv_min_waves(WM * WN, TM, TN, HALO)) void conv_igemm_kernel(ConvArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass only needs the launch stub (the buffer-resource builtins are device-only)
  constexpr int NW = WM * WN, NT = 64 * NW;           // waves / threads per block (4 or 8 waves)
  constexpr bool SPLIT = PREC != PREC_FP32;            // operands in split8 form (bf16 or fp16 pairs): 16 k per MFMA
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int AP = BM / 8 / NW, BP = BN / 8 / NW;   // 1 KiB (8-row) DMA pieces per wave and chunk
  constexpr int HR = conv_halo_rows(BM);              // HALO: LDS rows (pixels) of the resident image
  constexpr int ABUF = (HALO ? HR : BM) * LDS_STRIDE;
  constexpr int STAGE = ABUF + (BREG ? 0 : BN) * LDS_STRIDE;
  static_assert(!BREG || (HALO && SPLIT && BM * BN <= 2 * STAGE), "BREG needs the resident image");
  static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0, "pieces must divide over the waves");
  static_assert(!HALO || EPI != EPI_STORE, "the image layer is a 1-tap GEMM");
  static_assert(NS == 2 || (NS > 2 && !HALO && !BREG), "deeper staging exists for the plain (non-resident) A path only");
  constexpr int DPC = AP + BP;                         // DMA instructions per wave and chunk (plain staging)
  static_assert((NS - 2) * DPC <= 63, "vmcnt is a 6-bit counter");
  __shared__ __attribute__((aligned(16))) float smem[NS * STAGE];

  // ---- XCD-aware block remap: the n_tiles blocks that share an A tile get
  // consecutive logical ids and land on one XCD (one L2) [bijective form].
  const int nblk = a.m_tiles * a.n_tiles;
  int logical;
  {
    const int bid = blockIdx.x, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int mt = logical / a.n_tiles, nt = logical - mt * a.n_tiles;
  const int m0 = mt * BM, n0 = nt * BN;
  int pn = 0, py0 = 0, px0 = 0;                        // EPI_IMG_STENCIL: image slot and first OUTPUT pixel of the patch
  if constexpr (EPI == EPI_IMG_STENCIL) {
    const int tpi = a.tiles_x * a.tiles_y;
    pn = mt / tpi;
    const int r = mt - pn * tpi, ty = r / a.tiles_x;
    py0 = ty * IMG_TILE;
    px0 = (r - ty * a.tiles_x) * IMG_TILE;
  }
  int Y0 = 0, x0 = 0, img0 = 0;                          // HALO: first stack row / column of the tile, its image
  if constexpr (HALO) {
    const int mtp = a.tile_map ? a.tile_map[mt] : mt;    // launch order -> tile of the stack (TileOrder)
    const int tyt = mtp / a.cols_t;
    x0 = (mtp - tyt * a.cols_t) * a.tw;
    if (a.tpt > 0) {                                     // per-token tiling (ConvArgs::tpt)
      img0 = tyt / a.tpt;
      Y0 = img0 * a.H + (tyt - img0 * a.tpt) * a.th;
    } else {
      Y0 = tyt * a.th;
      img0 = Y0 / a.H;
    }
  }
  // exact small-integer division (operands < 2^22): float estimate + one fix-up step
  auto divmod = [](int x, int d, float inv, int& q, int& r) {
    q = (int)(((float)x + 0.5f) * inv);
    r = x - q * d;
    if (r < 0) { --q; r += d; } else if (r >= d) { ++q; r -= d; }
  };
  // first stack row that is NOT this tile's any more: the end of the stack, or of the tile's token when tiles are per token
  const int yend = HALO ? (a.tpt > 0 && (img0 + 1) * a.H < a.nyh ? (img0 + 1) * a.H : a.nyh) : 0;
  const float inv_tw = HALO ? 1.0f / (float)a.tw : 0.f, inv_H = HALO ? 1.0f / (float)a.H : 0.f;
  const float inv_H1 = HALO ? 1.0f / (float)(a.H + 1) : 0.f;
  // interleaved dual forward: the scale records of the (at most two, for all but tiny images) images this tile's rows belong
  // to are requested HERE, a main loop ahead of the epilogue that needs them — loaded there, indexed by the row's image,
  // their latency sat on every tile's critical path [MI355X: the forward convs 7 % slower]
  int simg0 = 0;
  float us0 = 1.f, us1 = 1.f, ps0 = 1.f, ps1 = 1.f;
  [[maybe_unused]] const float inv_img_rows = EPI == EPI_FWD_DUAL && a.scale_per_img ? 1.0f / (float)a.img_rows : 0.f;
  if constexpr (EPI == EPI_FWD_DUAL) {
    if (a.dual_il) {
      int simg1 = 0;
      if (a.scale_per_img) {
        simg0 = HALO ? img0 : m0 / a.img_rows;
        simg1 = simg0 + 1 < a.n_imgs ? simg0 + 1 : simg0;
      }
      if constexpr (PREC == PREC_F16X2) { us0 = a.in_unscale[simg0]; us1 = a.in_unscale[simg1]; }
      if (a.pairs_out) { ps0 = a.pairs_scale[simg0]; ps1 = a.pairs_scale[simg1]; }
    }
  }

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave - wm * WN;
  const int HW = a.H * a.W;
  const int K = a.taps * a.CinP;
  const int cpt = a.CinP >> 5;                         // 32-wide k chunks per tap
  const int nk = a.taps * cpt;

  // ---- staging: global -> LDS directly (buffer_load_dwordx4 ... lds), no VGPR round trip.
  // One wave instruction moves 64 x 16 B = 1 KiB = 8 staged rows; the LDS side is linear
  // (M0 base + lane*16), so the XOR swizzle is applied on the per-lane SOURCE offset: lane i of
  // piece q fills row 8q + (i>>3), physical chunk i&7, with logical chunk (i&7) ^ ((row>>1)&7).
  // Wave w issues pieces w*AP .. w*AP+AP-1 of the A tile and w*BP .. of the B tile.
  // Buffer addressing = SGPR base (the descriptor) + SGPR offset (tap / channel chunk: wave-uniform) + one
  // 32-bit VGPR offset per lane that never changes; out-of-image taps, ragged tails and padded channels set the
  // VGPR offset to a value beyond num_records and the hardware writes zeros — no pointer selects, no 64-bit
  // VALU adds, no branches in the loop.  Descriptors are rebased per block, so tensors > 4 GiB are fine.
  const int prow = lane >> 3, pchk = lane & 7;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  typedef __attribute__((address_space(3))) void* lptr_t;
  constexpr int OOB = (int)0x80000000;                 // >= any num_records used here
  constexpr int RSRC_FLAGS = 0x00020000;               // raw buffer, DATA_FORMAT_32 (gfx9 V# dword 3)
  const __amdgpu_buffer_rsrc_t rsB =
      __builtin_amdgcn_make_buffer_rsrc((void*)(a.wpk + (size_t)n0 * K), 0, BN * K * 4, RSRC_FLAGS);
  int bvo[BP];
#pragma unroll
  for (int p = 0; p < BP; ++p) {
    const int r = (wave_s * BP + p) * 8 + prow;
    bvo[p] = (r * K + ((pchk ^ ((r >> 1) & 7)) << 2)) * 4;
  }
  // A, non-HALO: descriptor base = pixel (m0 - W - 1), so that every tap offset is >= 0
  const float* abase = HALO ? a.in + (size_t)(Y0 > 0 ? Y0 - 1 : 0) * a.W * a.Cin
                       : EPI == EPI_IMG_STENCIL ? a.in + (size_t)pn * HW * a.Cin
                            : a.in + ((long)m0 - (a.taps == 1 ? 0 : a.W + 1)) * (long)a.Cin;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)abase, 0, 0x7FFFFFFF, RSRC_FLAGS);
  int avo[HALO ? 1 : AP];
  unsigned amask[HALO ? 1 : AP];
  int acf[HALO ? 1 : AP];                              // first channel (within a 32-chunk) of this lane's 16 B
  if constexpr (!HALO) {
    const int rfirst = wave_s * AP * 8 + prow;         // tile row of piece p = 0
    const int mfirst = m0 + rfirst;
    int rem = mfirst % HW;
    int h = rem / a.W, w = rem - h * a.W;
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      const int r = rfirst + 8 * p, m = m0 + r;
      unsigned mask = 0;
      if constexpr (EPI == EPI_IMG_STENCIL) {
        // patch row r = (py, px) -> pixel (py0 - 1 + py, px0 - 1 + px) of image slot pn; outside the image: zero row
        const int y = py0 - 1 + (r >> 4), x = px0 - 1 + (r & 15);
        const bool ok = y >= 0 && y < a.H && x >= 0 && x < a.W;
        amask[p] = ok ? 1u : 0u;
        const int lc = (pchk ^ ((r >> 1) & 7)) << 2;
        acf[p] = SPLIT ? ((lc >> 3) << 3) : lc;
        avo[p] = ok ? ((y * a.W + x) * a.Cin + lc) * 4 : 0;
        continue;
      }
      if (m < a.M) {
        if (a.taps == 1) {
          mask = 1u;
        } else {
          const unsigned row_ok = (h > 0 ? 0x007u : 0u) | 0x038u | (h < a.H - 1 ? 0x1C0u : 0u);   // taps 0-2 / 3-5 / 6-8
          const unsigned col_ok = (w > 0 ? 0x049u : 0u) | 0x092u | (w < a.W - 1 ? 0x124u : 0u);   // taps 0,3,6 / 1,4,7 / 2,5,8
          mask = row_ok & col_ok;
        }
      }
      amask[p] = mask;
      const int lc = (pchk ^ ((r >> 1) & 7)) << 2;
      acf[p] = SPLIT ? ((lc >> 3) << 3) : lc;   // 4*chunk (fp32) or 8*(chunk/2) (split8 group)
      avo[p] = (r * a.Cin + lc) * 4;
      if (a.taps != 1) {                                  // next piece: 8 pixels further
        w += 8;
        while (w >= a.W) { w -= a.W; ++h; }
        while (h >= a.H) h -= a.H;
      }
    }
  }

  int tap = 0, cc = 0;                                 // position of the chunk being LOADED
  // HALO: 8-row piece p of the resident image of channel chunk `chunk` -> A buffer `abuf`.
  // piece p = half of resident row hy (wave-uniform): extended stack row E -> image n, row h (h == H: separator)
  const int halo_np = HALO ? a.hrows * (HALO_PITCH / 8) : 0;
  struct HaloPiece { int vo, so; };
  auto prep_halo_piece = [&](int p, int chunk, bool live) {
    const int hy = p >> 1, hx = (p & 1) * 8 + prow;
    const int E = Y0 + img0 - 1 + hy;
    int n, h;
    divmod(E < 0 ? 0 : E, a.H + 1, inv_H1, n, h);
    const int x = x0 - 1 + hx;
    const int lc = (pchk ^ (((hy * a.tw + hx - 1) >> 1) & 7)) << 2;   // halo swizzle, see set_tap
    const int cfirst = SPLIT ? ((lc >> 3) << 3) : lc;
    const bool ok = live && E >= 0 && h < a.H && n < a.NB && hx < a.tw + 2 && x >= 0 && x < a.W && (chunk << 5) + cfirst < a.Cin;
    HaloPiece r;
    r.vo = ok ? (x * a.Cin + lc) * 4 : OOB;
    // (n comes out of float math = VALU; the offset is wave-uniform and must reach the instruction in an SGPR)
    const int so = (((E - n) - (Y0 > 0 ? Y0 - 1 : 0)) * a.W * a.Cin + (chunk << 5)) * 4;
    r.so = __builtin_amdgcn_readfirstlane(so < 0 ? 0 : so);
    return r;
  };
  auto fire_halo_piece = [&](const HaloPiece& hp, int p, int abuf) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lptr_t)(smem + abuf * STAGE + p * 8 * LDS_STRIDE), 16, hp.vo, hp.so, 0, 0);
  };
  // prep: per-lane offsets of the chunk at (tap, cc) (cheap VALU, placed BEFORE the barrier where it overlaps the
  // MFMAs still in flight); fire: the DMA instructions themselves, right after the barrier.
  struct ChunkPrep { int avo[HALO ? 1 : AP]; int aso, bso; };
  auto prep_chunk = [&](bool live) {
    ChunkPrep c;
    const int c0 = cc << 5;
    if constexpr (!HALO) {
#pragma unroll
      for (int p = 0; p < AP; ++p) {
        const bool ok = live && ((amask[p] >> tap) & 1u) && (c0 + acf[p] < a.Cin);
        c.avo[p] = ok ? avo[p] : OOB;
      }
    }
    c.aso = ((a.taps == 1 ? 0 : (tap / 3) * a.W + (tap % 3)) * a.Cin + c0) * 4;
    c.bso = live ? (tap * a.CinP + c0) * 4 : 0;
    // taps innermost: the 9 taps of one 32-channel chunk touch the same ~(BM + halo) pixel rows
    // (24 KB), so 8 of 9 re-reads hit L1/L2; tap-major order streamed BM x Cin x 4 B per tap
    // through a 64 KB-per-block share of the XCD's L2 and missed on every tap (FETCH_SIZE 7x).
    if (++tap == a.taps) { tap = 0; ++cc; }
    return c;
  };
  auto fire_chunk = [&](const ChunkPrep& c, int buf) {
    float* As = smem + buf * STAGE;
    float* Bs = As + ABUF;
    if constexpr (!HALO) {
#pragma unroll
      for (int p = 0; p < AP; ++p)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lptr_t)(As + (wave_s * AP + p) * 8 * LDS_STRIDE), 16, c.avo[p], c.aso, 0, 0);
    }
#pragma unroll
    for (int p = 0; p < BP; ++p)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lptr_t)(Bs + (wave_s * BP + p) * 8 * LDS_STRIDE), 16, bvo[p], c.bso, 0, 0);
  };

  // Two-level (blocked) summation: the MFMA is a strictly k-ordered fp32 fma chain, so a
  // K = 4608 reduction in ONE accumulator carries ~sqrt(K) ulp of round-off.  Every FLUSH
  // chunks (256 k) the running block is folded into `tot` and restarted: chains of 256 + K/256.
  // (bf16x3: one MFMA already folds 16 k internally and the chain is K/16 long — no second level.)
  constexpr int FLUSH = 8;
  // (the four-term pass of the exact forward product rounds its accumulator 4 x K/16 times: blocked as well — measured
  //  feature error of VGG16 2.3e-6 without)
  // (and the fp16-pair forward, whose three-term product is fp32-grade: without the second level its accumulator's
  //  3 x K/16 roundings would be the largest error left)
  constexpr bool BLOCKED = PREC == PREC_FP32 || TERMS == 15 ||
                           (PREC == PREC_F16X2 && (EPI == EPI_BIAS || EPI == EPI_BIAS_RELU || EPI == EPI_FWD_DUAL));
  f32x16 acc[TM][TN], tot[BLOCKED ? TM : 1][BLOCKED ? TN : 1];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[i][j][r] = 0.f;
        if constexpr (BLOCKED) tot[i][j][r] = 0.f;
      }

  // DMA schedule: chunk kc+2 is launched right AFTER the barrier of iteration kc (the barrier proves every wave
  // has finished reading buffer kc&1) and is waited for at the barrier of iteration kc+1, so a load has a whole
  // iteration (48 MFMAs per wave on the 8-wave tile) to land — launching it at the top of iteration kc+1 instead
  // left it half of that, which the HBM / MALL latency of the A rows did not fit into in the bf16x3 mode.
  // ---- compact pool interface in the PIPELINED halo kernels (!BREG; ConvArgs::up2_pairs): the producer has written
  // S_c = acc x (compact gate) at POOLED resolution as bf16 pairs, and the resident image of the next channel chunk is built
  // from (S_c, position bytes) through registers while the current chunk's taps run, instead of arriving by DMA from the
  // 4x-expanded, 75 %-zero tensor.  Item = (resident row, window column, 8-channel group): it loads its window's pairs
  // (32 B) and position bytes (8 B) and writes the row's two pixels of that window — the value where the position matches,
  // zeros elsewhere (and zeros for separator rows / rows outside the stack).  At most two items per thread and chunk (the
  // launcher checks), handled one after the other so that only ten registers are in flight: item 0 is loaded right after
  // the barrier of tap 3 and written before the barrier of tap 5, item 1 after tap 5 / before tap 7; every write is two
  // barriers ahead of the chunk's first read, into the A buffer last read a chunk ago.  Works on stack tiles and per-token tiles.
  typedef unsigned u32x2w __attribute__((ext_vector_type(2)));
  struct PwItem { u32x4 hi, lo; u32x2w q; };
  [[maybe_unused]] const bool pw = HALO && !BREG && PREC == PREC_BF16X3 && a.up2_src != nullptr;
  [[maybe_unused]] int pw_wx0 = 0, pw_nwx = 1, pw_items = 0, pw_ri0 = 0, pw_ri1 = 0;
  [[maybe_unused]] float pw_inv_nwx = 1.f;
  if constexpr (HALO && !BREG && PREC == PREC_BF16X3) {
    if (pw) {
      pw_wx0 = (x0 - 1) >> 1;                           // (arithmetic shift: the window column outside the image for x0 = 0)
      pw_nwx = ((x0 + a.tw) >> 1) - pw_wx0 + 1;
      pw_inv_nwx = 1.0f / (float)pw_nwx;
      pw_items = a.hrows * pw_nwx * 4;
      pw_ri0 = a.row2img ? a.row2img[img0 < a.NB ? img0 : a.NB - 1] : img0;
      pw_ri1 = a.row2img ? a.row2img[img0 + 1 < a.NB ? img0 + 1 : a.NB - 1] : img0 + 1;
    }
  }
  struct PwPos { const float* sp; const unsigned char* qp; bool valid; int hy, wx, g, dy; };
  [[maybe_unused]] auto pw_locate = [&](int it, int chunk) {
    PwPos r;
    r.g = it & 3;
    int wxi;
    divmod(it >> 2, pw_nwx, pw_inv_nwx, r.hy, wxi);
    r.wx = pw_wx0 + wxi;
    const int Hp = a.H >> 1, Wp = a.W >> 1;
    const int E = Y0 + img0 - 1 + r.hy;
    int n, h;
    divmod(E < 0 ? 0 : E, a.H + 1, inv_H1, n, h);
    r.valid = it < pw_items && E >= 0 && h < a.H && n < a.NB && r.wx >= 0 && r.wx < Wp && (chunk << 5) + r.g * 8 < a.Cin;
    r.dy = h & 1;
    // unconditional loads from a clamped (always valid) place; an invalid item's values are zeroed at the store
    const int nc = r.valid ? n : 0, wyc = r.valid ? (h >> 1) : 0, wxc = r.valid ? r.wx : 0, co = r.valid ? (chunk << 5) + r.g * 8 : 0;
    const int rel = nc - img0;
    const int img = rel == 0 ? pw_ri0 : rel == 1 ? pw_ri1 : (a.row2img ? a.row2img[nc] : nc);
    r.sp = a.up2_src + (((size_t)nc * Hp + wyc) * Wp + wxc) * a.Cin + co;
    r.qp = a.up2_gpos + (((size_t)img * Hp + wyc) * Wp + wxc) * a.Cin + co;
    return r;
  };
  [[maybe_unused]] auto pw_load = [&](PwItem& w, int it, int chunk) {
    const PwPos r = pw_locate(it, chunk);
    w.hi = *reinterpret_cast<const u32x4*>(r.sp);
    w.lo = *reinterpret_cast<const u32x4*>(r.sp + 4);
    w.q = *reinterpret_cast<const u32x2w*>(r.qp);
  };
  [[maybe_unused]] auto pw_store = [&](const PwItem& w, int it, int chunk, int abuf) {
    if (it >= pw_items) return;
    const PwPos r = pw_locate(it, chunk);
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      const int hx = 2 * r.wx + dx - (x0 - 1);
      if (hx < 0 || hx >= HALO_PITCH) continue;
      const unsigned j = (unsigned)((r.dy << 1) | dx);  // this pixel's position in its window
      u32x4 mh, ml;
#pragma unroll
      for (int d = 0; d < 4; ++d) {                       // 16-bit lane c keeps its value iff channel c's arg-max sits at position j
        const unsigned p0 = (w.q[d >> 1] >> (16 * (d & 1))) & 0xFFu, p1 = (w.q[d >> 1] >> (16 * (d & 1) + 8)) & 0xFFu;
        const unsigned m = r.valid ? ((p0 == j ? 0x0000FFFFu : 0u) | (p1 == j ? 0xFFFF0000u : 0u)) : 0u;
        mh[d] = w.hi[d] & m;
        ml[d] = w.lo[d] & m;
      }
      const int row = r.hy * HALO_PITCH + hx;
      const int swzu = ((r.hy * a.tw + hx - 1) >> 1) & 7;
      const int dst = abuf * STAGE + row * LDS_STRIDE + (((2 * r.g) ^ swzu) << 2);
      *reinterpret_cast<u32x4*>(smem + dst) = mh;
      *reinterpret_cast<u32x4*>(smem + (dst ^ 4)) = ml;
    }
  };
  [[maybe_unused]] PwItem pwi{};
  if constexpr (BREG) {
    if (PREC == PREC_BF16X3 && a.up2_src && (a.up2_gc || a.up2_pairs) && a.up2_gpos && a.tpt > 0) {
      // compact pool interface, compact gate: item = (window of the resident image, 8-channel group)
      const bool pairs = a.up2_pairs != 0;
      const int Hp = a.H >> 1, Wp = a.W >> 1;
      const int h0 = Y0 - img0 * a.H;                      // the tile's first image row (tiles are per token)
      const int wy0 = (h0 - 1) >> 1, wx0 = (x0 - 1) >> 1;  // (arithmetic shifts: -1 >> 1 = -1, the window row / column outside the image)
      const int nwx = ((x0 + a.tw) >> 1) - wx0 + 1, nwy = ((h0 + a.th) >> 1) - wy0 + 1;
      const int items = nwy * nwx * 8;
      const int ntok = img0 < a.NB ? img0 : a.NB - 1;
      const int img = a.row2img ? a.row2img[ntok] : ntok;
      const float inv_nwx = 1.0f / (float)nwx;
      constexpr int UW = 2;                                // items in flight per thread (4 x 16 B + 8 B of loads each)
      typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
      for (int it0 = tid; it0 < items; it0 += NT * UW) {
        f32x4 pv[UW][2], gv[UW][2] = {};
        u32x2_ qv[UW];
        int wyv[UW], wxv[UW], cgv[UW];
        bool okv[UW];
#pragma unroll
        for (int u = 0; u < UW; ++u) {
          const int item = it0 + u * NT;
          const int cg = item & 7, wdx = item >> 3;
          int wr, wc;
          divmod(wdx, nwx, inv_nwx, wr, wc);
          const int wy = wy0 + wr, wx = wx0 + wc;
          cgv[u] = cg; wyv[u] = wy; wxv[u] = wx;
          okv[u] = item < items && wy >= 0 && wy < Hp && wx >= 0 && wx < Wp && cg * 8 < a.Cin;
          // unconditional loads from a clamped (always valid) window; the values of an invalid item are zeroed below
          const int wyc = wy < 0 ? 0 : (wy >= Hp ? Hp - 1 : wy), wxc = wx < 0 ? 0 : (wx >= Wp ? Wp - 1 : wx);
          const int cgc = cg * 8 < a.Cin ? cg : 0;
          const size_t wo = ((size_t)wyc * Wp + wxc) * a.Cin + cgc * 8;
          const float* pp = a.up2_src + (size_t)ntok * Hp * Wp * a.Cin + wo;
          pv[u][0] = *reinterpret_cast<const f32x4*>(pp); pv[u][1] = *reinterpret_cast<const f32x4*>(pp + 4);
          if (!pairs) {                                   // (block-uniform; pairs mode: up2_src already holds P x gate as [hi8 | lo8])
            const float* gp = a.up2_gc + (size_t)img * Hp * Wp * a.Cin + wo;
            gv[u][0] = *reinterpret_cast<const f32x4*>(gp); gv[u][1] = *reinterpret_cast<const f32x4*>(gp + 4);
          }
          qv[u] = *reinterpret_cast<const u32x2_*>(a.up2_gpos + (size_t)img * Hp * Wp * a.Cin + wo);
        }
#pragma unroll
        for (int u = 0; u < UW; ++u) {
          if (it0 + u * NT >= items) continue;
          u32x4 hiw, low;
          if (pairs) {
            const u32x4 z4u = {0u, 0u, 0u, 0u};
            hiw = okv[u] ? __builtin_bit_cast(u32x4, pv[u][0]) : z4u;
            low = okv[u] ? __builtin_bit_cast(u32x4, pv[u][1]) : z4u;
          } else {
            float r[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { r[e] = pv[u][0][e] * gv[u][0][e]; r[4 + e] = pv[u][1][e] * gv[u][1][e]; }
            bf16x8 hi, lo;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              const float rq = okv[u] ? r[q] : 0.f;
              hi[q] = (__bf16)rq;
              lo[q] = (__bf16)(rq - (float)hi[q]);
            }
            hiw = __builtin_bit_cast(u32x4, hi); low = __builtin_bit_cast(u32x4, lo);
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {                    // the window's four pixels: position j = 2 dy + dx
            const int hy = 2 * wyv[u] + (j >> 1) - (h0 - 1), hx = 2 * wxv[u] + (j & 1) - (x0 - 1);
            if (hy < 0 || hy >= a.hrows || hx < 0 || hx >= HALO_PITCH) continue;
            // 16-bit lane c of a dword pair keeps its value iff channel c's arg-max sits at position j
            u32x4 mh, ml;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
              const unsigned p0 = (qv[u][d >> 1] >> (16 * (d & 1))) & 0xFFu, p1 = (qv[u][d >> 1] >> (16 * (d & 1) + 8)) & 0xFFu;
              const unsigned m = (p0 == (unsigned)j ? 0x0000FFFFu : 0u) | (p1 == (unsigned)j ? 0xFFFF0000u : 0u);
              mh[d] = hiw[d] & m;
              ml[d] = low[d] & m;
            }
            const int row = hy * HALO_PITCH + hx;
            const int swzu = ((hy * a.tw + hx - 1) >> 1) & 7;
            const int dst = (cgv[u] >> 2) * STAGE + row * LDS_STRIDE + (((2 * (cgv[u] & 3)) ^ swzu) << 2);
            *reinterpret_cast<u32x4*>(smem + dst) = mh;
            *reinterpret_cast<u32x4*>(smem + (dst ^ 4)) = ml;
          }
        }
      }
    } else if (PREC == PREC_BF16X3 && a.up2_src) {
      // compact pool interface: resident image = P (pooled resolution) x pool gate, built here through registers.
      // item = (LDS row, 8-channel group of the 64 channels); chunk cc = group / 4 goes to LDS buffer cc.
      const int Hp = a.H >> 1, Wp = a.W >> 1;
      const int items = a.hrows * HALO_PITCH * 8;
      const int ri0 = a.row2img ? a.row2img[img0 < a.NB ? img0 : a.NB - 1] : img0;
      const int ri1 = a.row2img ? a.row2img[img0 + 1 < a.NB ? img0 + 1 : a.NB - 1] : img0 + 1;
      constexpr int UB = 6;                              // items in flight per thread (4 x 16 B loads each): one round for the 128-row tile
      for (int it0 = tid; it0 < items; it0 += NT * UB) {
        f32x4 pv[UB][2], gv[UB][2];
        int dst[UB];
        bool okv[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          const int item = it0 + u * NT;
          const int row = item >> 3, cg = item & 7, hy = row / HALO_PITCH, hx = row - hy * HALO_PITCH;
          const int E = Y0 + img0 - 1 + hy;
          int n, h;
          divmod(E < 0 ? 0 : E, a.H + 1, inv_H1, n, h);
          const int x = x0 - 1 + hx;
          okv[u] = item < items && E >= 0 && h < a.H && n < a.NB && hx < a.tw + 2 && x >= 0 && x < a.W && cg * 8 < a.Cin;
          const int swzu = ((hy * a.tw + hx - 1) >> 1) & 7;
          // hi chunk of group g = cg & 3 is logical chunk 2g, lo chunk 2g + 1; physical = logical ^ swizzle (see set_tap)
          dst[u] = (cg >> 2) * STAGE + row * LDS_STRIDE + (((2 * (cg & 3)) ^ swzu) << 2);
          const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
          pv[u][0] = pv[u][1] = gv[u][0] = gv[u][1] = z4;
          if (okv[u]) {
            const int rel = n - img0;
            const int img = rel == 0 ? ri0 : rel == 1 ? ri1 : (a.row2img ? a.row2img[n] : n);
            const float* pp = a.up2_src + (((size_t)n * Hp + (h >> 1)) * Wp + (x >> 1)) * a.Cin + cg * 8;
            const float* gp = a.up2_gate + (((size_t)img * a.H + h) * a.W + x) * a.Cin + cg * 8;
            pv[u][0] = *reinterpret_cast<const f32x4*>(pp); pv[u][1] = *reinterpret_cast<const f32x4*>(pp + 4);
            gv[u][0] = *reinterpret_cast<const f32x4*>(gp); gv[u][1] = *reinterpret_cast<const f32x4*>(gp + 4);
          }
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          if (it0 + u * NT >= items) continue;
          float r[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) { r[e] = pv[u][0][e] * gv[u][0][e]; r[4 + e] = pv[u][1][e] * gv[u][1][e]; }
          bf16x8 hi, lo;
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            hi[q] = (__bf16)r[q];
            lo[q] = (__bf16)(r[q] - (float)hi[q]);
          }
          // lo sits in the logical chunk next to hi: physical index differs in bit 0 only (the swizzle XORs whole indices)
          *reinterpret_cast<u32x4*>(smem + dst[u]) = __builtin_bit_cast(u32x4, hi);
          *reinterpret_cast<u32x4*>(smem + (dst[u] ^ 4)) = __builtin_bit_cast(u32x4, lo);
        }
      }
    } else
    for (int p = wave_s; p < halo_np; p += NW) {         // channel chunks 0 and 1 -> LDS buffers 0 and 1
      fire_halo_piece(prep_halo_piece(p, 0, true), p, 0);
      fire_halo_piece(prep_halo_piece(p, 1, cpt > 1), p, 1);
    }
  } else {
    if constexpr (HALO) {
      if (pw) {
        pw_load(pwi, tid, 0); pw_store(pwi, tid, 0, 0);
        if (tid + NT < pw_items) { pw_load(pwi, tid + NT, 0); pw_store(pwi, tid + NT, 0, 0); }
      } else {
        for (int p = wave_s; p < halo_np; p += NW) fire_halo_piece(prep_halo_piece(p, 0, true), p, 0);
        if (wave_s < halo_np) fire_halo_piece(prep_halo_piece(wave_s, 1, cpt > 1), wave_s, 1);   // slot 0 of chunk 1 (see the loop)
      }
    }
    fire_chunk(prep_chunk(true), 0);
    fire_chunk(prep_chunk(nk > 1), 1);
#pragma unroll
    for (int s_ = 2; s_ < NS; ++s_) fire_chunk(prep_chunk(nk > s_), s_);
  }
  // hipcc gives __syncthreads() an lgkmcnt(0) only; the LDS-DMA completes on vmcnt, so the wait is explicit
  auto dma_landed_barrier = [] {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };
  // NS > 2: the oldest chunk in flight has landed (the NS - 2 younger ones may still be on their way)
  auto dma_oldest_landed_barrier = [] {
    if constexpr (NS > 2) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * ((WM * TM * 32 + WN * TN * 32) / 8 / (WM * WN))) : "memory");   // (NS - 2) x DPC
      __syncthreads();
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  };
  dma_oldest_landed_barrier();

  const int a_off = HALO ? 0 : (wm * TM * 32 + (lane & 31)) * LDS_STRIDE;
  const int b_off = ABUF + (wn * TN * 32 + (lane & 31)) * LDS_STRIDE;
  // HALO: per A fragment row, the LDS row of its pixel in the resident image (padding rows of the tile read
  // pixel (1,1): their C rows are never stored), and per tap the float offset of its first 16 B chunk
  int fbase[HALO ? TM : 1], fu[HALO ? TM : 1], faddr[HALO ? TM : 1];
  const int hh_ = lane >> 5;
  if constexpr (HALO) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int r = (wm * TM + i) * 32 + (lane & 31);
      int ty, tx, n_, h_;
      divmod(r, a.tw, inv_tw, ty, tx);
      const int Y = Y0 + ty;
      divmod(Y, a.H, inv_H, n_, h_);
      const bool ok = r < a.th * a.tw && Y < yend && x0 + tx < a.W;
      const int hy = ok ? ty + 1 + n_ - img0 : 1, hx = ok ? tx + 1 : 1;
      fbase[i] = hy * HALO_PITCH + hx;
      fu[i] = hy * a.tw + hx - 1;
    }
  }
  auto tap_shift = [&](int t) { return (t / 3 - 1) * HALO_PITCH + (t % 3 - 1); };
  // first chunk of lane-half hh at step 0 is 2hh (bf16x3: hi of split8 group hh) / hh (fp32); the others are XORs of it
  // Halo swizzle: the 16 B chunk index is XORed with (u >> 1) & 7, u = hy*tw + hx - 1 = the pixel's index in the
  // DENSE tile (the LDS rows have pitch 16 but only tw = 14 of them per image row are tile pixels).  The 32 lanes
  // of a fragment read 32 consecutive tile pixels, i.e. consecutive u (shifted as a whole by the tap), and 16
  // consecutive u give 16 distinct (row parity, chunk ^ swizzle) bank slots; the LDS-row based swizzle of the
  // non-halo layout would collide on the two rows behind every wrap of tx.
  auto set_tap = [&](int t) {
    const int tsh = tap_shift(t), ush = (t / 3 - 1) * a.tw + (t % 3 - 1);
#pragma unroll
    for (int i = 0; i < (HALO ? TM : 0); ++i) {
      const int row = fbase[i] + tsh, u = fu[i] + ush;
      faddr[i] = row * LDS_STRIDE + ((((SPLIT ? 2 : 1) * hh_) ^ ((u >> 1) & 7)) << 2);
    }
  };
  // Fragment registers are double-buffered one step ahead, and the first fragments of the NEXT
  // chunk are fetched right after the barrier: the last MFMA group of the current chunk runs on
  // registers while those reads are in flight.  (All reads of `buf` are issued AND completed
  // before the barrier => the DMA of the following iteration cannot race them.)
  // fp32 : 4 steps of 8 k per chunk; lane-half h consumes k = 4h+s -> logical chunk 2kk+h.
  // bf16x3: 2 steps of 16 k; lane-half h consumes k = 8h+j -> split8 group 2s+h = chunks 4s+2h (hi), 4s+2h+1 (lo).
  constexpr int NSTEP = SPLIT ? 2 : 4;
  const int swz = (lane >> 1) & 7, hh = lane >> 5;
  int koff[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
    koff[q] = SPLIT ? (((4 * (q >> 1) + 2 * hh + (q & 1)) ^ swz) << 2)     // q = 2*step + {hi,lo}
                                  : (((2 * q + hh) ^ swz) << 2);
  struct Frag { u32x4 a[SPLIT ? 2 * TM : TM], b[SPLIT ? 2 * TN : TN]; };
  // HALO: A addresses come from faddr[] (set_tap); chunk c ^ swizzle = (c0 ^ swizzle) ^ (c - c0) for disjoint bits
  auto read_frag = [&](Frag& f, const float* Ab, const float* Bb, int st) {
    if constexpr (HALO) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if constexpr (SPLIT) {
          f.a[2 * i] = *reinterpret_cast<const u32x4*>(Ab + (faddr[i] ^ ((4 * st) << 2)));
          f.a[2 * i + 1] = *reinterpret_cast<const u32x4*>(Ab + (faddr[i] ^ ((4 * st + 1) << 2)));
        } else {
          f.a[i] = *reinterpret_cast<const u32x4*>(Ab + (faddr[i] ^ ((2 * st) << 2)));
        }
      }
    } else if constexpr (SPLIT) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        f.a[2 * i] = *reinterpret_cast<const u32x4*>(Ab + i * 32 * LDS_STRIDE + koff[2 * st]);
        f.a[2 * i + 1] = *reinterpret_cast<const u32x4*>(Ab + i * 32 * LDS_STRIDE + koff[2 * st + 1]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < TM; ++i) f.a[i] = *reinterpret_cast<const u32x4*>(Ab + i * 32 * LDS_STRIDE + koff[st]);
    }
    if constexpr (SPLIT) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        f.b[2 * j] = *reinterpret_cast<const u32x4*>(Bb + j * 32 * LDS_STRIDE + koff[2 * st]);
        if (PREC != PREC_F16X2 || ((TERMS & 2) != 0 && !((TERMS & 16) != 0 && (j & 1))))   // (two-term form: the lo chunk is never read)
          f.b[2 * j + 1] = *reinterpret_cast<const u32x4*>(Bb + j * 32 * LDS_STRIDE + koff[2 * st + 1]);
      }
    } else {
#pragma unroll
      for (int j = 0; j < TN; ++j) f.b[j] = *reinterpret_cast<const u32x4*>(Bb + j * 32 * LDS_STRIDE + koff[st]);
    }
  };
  auto mfma_frag = [&](const Frag& f) {
    if constexpr (PREC == PREC_F16X2) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const f16x8 ah = __builtin_bit_cast(f16x8, f.a[2 * i]), al = __builtin_bit_cast(f16x8, f.a[2 * i + 1]);
          const f16x8 bh = __builtin_bit_cast(f16x8, f.b[2 * j]);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[i][j], 0, 0, 0);    // small terms first
          // TERMS 7: the weights' lo half too.  TERMS bit 4 (interleaved dual forward): not for the odd column tiles = Z+
          if ((TERMS & 2) != 0 && !((TERMS & 16) != 0 && (j & 1)))          // (folds once the j loop is unrolled)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, __builtin_bit_cast(f16x8, f.b[2 * j + 1]), acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[i][j], 0, 0, 0);
        }
    } else if constexpr (PREC == PREC_BF16X3) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const bf16x8 ah = __builtin_bit_cast(bf16x8, f.a[2 * i]), al = __builtin_bit_cast(bf16x8, f.a[2 * i + 1]);
          const bf16x8 bh = __builtin_bit_cast(bf16x8, f.b[2 * j]), bl = __builtin_bit_cast(bf16x8, f.b[2 * j + 1]);
          if constexpr ((TERMS & 8) != 0) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bl, acc[i][j], 0, 0, 0);
          if constexpr ((TERMS & 1) != 0) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[i][j], 0, 0, 0);    // small terms first
          if constexpr ((TERMS & 2) != 0) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[i][j], 0, 0, 0);
          if constexpr ((TERMS & 4) != 0) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[i][j], 0, 0, 0);
        }
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(f32x4, f.a[i])[s],
                                                             __builtin_bit_cast(f32x4, f.b[j])[s], acc[i][j], 0, 0, 0);
    }
  };

  if constexpr (BREG) {
    static_assert(NSTEP == 2, "bf16x3");
    // B fragments of (kc = chunk*9 + tap): [step][hi|lo] x TN, one 16 B load each
    const __amdgpu_buffer_rsrc_t rsF = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpk_frag, 0, nk * 8 * BN * 16, RSRC_FLAGS);
    const int fvo = ((lane >> 5) * BN + wn * TN * 32 + (lane & 31)) * 16;
    struct BFrag { u32x4 v[4 * TN]; };
    auto load_b = [&](BFrag& b, int kc) {
#pragma unroll
      for (int q = 0; q < 4; q += ((PREC == PREC_F16X2 && !(TERMS & 2)) ? 2 : 1))    // q = 2*step + (0 hi | 1 lo); two-term fp16: hi only
#pragma unroll
        for (int j = 0; j < TN; ++j)
          b.v[q * TN + j] = __builtin_amdgcn_raw_buffer_load_b128(rsF, fvo + j * 32 * 16, ((kc * 4 + q) * 2) * BN * 16, 0);
    };
    struct AFrag { u32x4 v[4 * TM]; };                   // [step][hi|lo] x TM
    auto load_a = [&](AFrag& f, const float* Ab) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < TM; ++i)
          f.v[q * TM + i] = *reinterpret_cast<const u32x4*>(Ab + (faddr[i] ^ (((q >> 1) * 4 + (q & 1)) << 2)));
    };
    auto mma = [&](const AFrag& f, const BFrag& b) {
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            if constexpr (PREC == PREC_F16X2) {
              const f16x8 ah = __builtin_bit_cast(f16x8, f.v[(2 * st) * TM + i]), al = __builtin_bit_cast(f16x8, f.v[(2 * st + 1) * TM + i]);
              const f16x8 bh = __builtin_bit_cast(f16x8, b.v[(2 * st) * TN + j]);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[i][j], 0, 0, 0);
              if constexpr ((TERMS & 2) != 0)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, __builtin_bit_cast(f16x8, b.v[(2 * st + 1) * TN + j]), acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[i][j], 0, 0, 0);
            } else {
              const bf16x8 ah = __builtin_bit_cast(bf16x8, f.v[(2 * st) * TM + i]), al = __builtin_bit_cast(bf16x8, f.v[(2 * st + 1) * TM + i]);
              const bf16x8 bh = __builtin_bit_cast(bf16x8, b.v[(2 * st) * TN + j]), bl = __builtin_bit_cast(bf16x8, b.v[(2 * st + 1) * TN + j]);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[i][j], 0, 0, 0);
            }
          }
    };
    BFrag b0, b1;
    AFrag a0, a1;
    load_b(b0, 0);
    for (int g = 0; g < cpt; g += 2) {                   // groups of two channel chunks = the two LDS buffers
      if (g > 0) {
        __syncthreads();                                 // everyone is done with the previous group's image
        for (int p = wave_s; p < halo_np; p += NW) {
          fire_halo_piece(prep_halo_piece(p, g, true), p, 0);
          fire_halo_piece(prep_halo_piece(p, g + 1, g + 1 < cpt), p, 1);
        }
        dma_landed_barrier();
      }
      const int ng = (g + 2 <= cpt ? 2 : 1) * 9;         // (tap, chunk) pairs of this group
      set_tap(0);
      load_a(a0, smem);
      // Two pairs per trip (the register double buffers alternate).  The body is branch-free on purpose — prefetches
      // past the end are clamped re-loads — so that it stays ONE basic block and hipcc can wait for exactly the four
      // older weight loads (vmcnt(4)); with branches around the loads it fell back to vmcnt(0) on every tap.
      int nt = 0, nc = 0;                                // (tap, chunk in group) of the pair being prefetched
      for (int q = 0; q < ng; q += 2) {
        const int kc = g * 9 + q;
        if (++nt == 9) { nt = 0; ++nc; }
        load_b(b1, kc + 1 < nk ? kc + 1 : nk - 1);
        set_tap(nt);
        load_a(a1, smem + (nc > 1 ? 1 : nc) * STAGE);
        __builtin_amdgcn_sched_barrier(0);               // keep the prefetches ABOVE the 12 MFMAs they hide behind
        mma(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        if (++nt == 9) { nt = 0; ++nc; }
        load_b(b0, kc + 2 < nk ? kc + 2 : nk - 1);
        set_tap(nt);
        load_a(a0, smem + (nc > 1 ? 1 : nc) * STAGE);
        __builtin_amdgcn_sched_barrier(0);
        if (q + 1 < ng) mma(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  Frag f0, f1;
  int ctap = 0, ccc = 0;                               // HALO: (tap, channel chunk) being COMPUTED; kc = ccc*9 + ctap
  if constexpr (!BREG) {
  set_tap(0);
  read_frag(f0, smem + a_off, smem + b_off, 0);
  }
  int buf = 0;                                         // = kc % NS
  for (int kc = 0; kc < (BREG ? 0 : nk); ++kc) {
    const bool more = (kc + 1) < nk;
    const int nbuf = buf + 1 == NS ? 0 : buf + 1;      // stage of chunk kc + 1
    const int abuf = HALO ? (ccc & 1) : buf;
    const float* Ab = smem + abuf * STAGE + a_off;
    const float* Bb = smem + buf * STAGE + b_off;
    if constexpr (NSTEP == 4) {
      read_frag(f1, Ab, Bb, 1);
      mfma_frag(f0);                                   // step 0
      read_frag(f0, Ab, Bb, 2);
      mfma_frag(f1);                                   // step 1
      read_frag(f1, Ab, Bb, 3);
      mfma_frag(f0);                                   // step 2
    } else {
      read_frag(f1, Ab, Bb, 1);
      mfma_frag(f0);                                   // step 0
    }
    // offsets of chunk kc+2 (-> `buf`) and, HALO, of this tap slot's piece of the next channel chunk's resident
    // image (slot = one piece per wave; that A buffer was last read in the chunk before this one)
    const ChunkPrep cp = prep_chunk(kc + NS < nk);
    HaloPiece hp{};
    int hpp = 0;
    if constexpr (HALO) {
      if (++ctap == 9) { ctap = 0; ++ccc; }
      if (pw) {
        hpp = -1;
        if (ccc + 1 < cpt) {
          if (ctap == 5) pw_store(pwi, tid, ccc + 1, (ccc & 1) ^ 1);
          else if (ctap == 7) pw_store(pwi, tid + NT, ccc + 1, (ccc & 1) ^ 1);
        }
      } else {
        hpp = ctap * NW + wave_s;
        if (hpp >= halo_np) hpp = -1;
        hp = prep_halo_piece(hpp < 0 ? 0 : hpp, ccc + 1, more && ccc + 1 < cpt);
      }
    }
    dma_oldest_landed_barrier();                       // reads of `buf` done everywhere; DMA of chunk kc+1 landed
    fire_chunk(cp, buf);
    if constexpr (HALO) {
      if (pw) {
        if (ccc + 1 < cpt) {
          if (ctap == 3) pw_load(pwi, tid, ccc + 1);
          else if (ctap == 5 && tid + NT < pw_items) pw_load(pwi, tid + NT, ccc + 1);
        }
      } else if (hpp >= 0 && more && ccc + 1 < cpt) fire_halo_piece(hp, hpp, (ccc & 1) ^ 1);
    }
    // keep the MFMAs below the DMA launch: without this the compiler hoists all of them above it (they do not
    // depend on it) and the loads start ~24 MFMA issue slots later
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(acc[i][j]));
    // (on the last iteration this reads a stale buffer into registers nobody uses)
    if constexpr (HALO) {
      set_tap(ctap);
      read_frag(f0, smem + (ccc & 1) * STAGE, smem + nbuf * STAGE + b_off, 0);
    } else {
      read_frag(f0, smem + nbuf * STAGE + a_off, smem + nbuf * STAGE + b_off, 0);
    }
    mfma_frag(f1);                                     // last step
    if constexpr (BLOCKED) {
      if ((kc & (FLUSH - 1)) == FLUSH - 1) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            tot[i][j] += acc[i][j];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
          }
      }
    }
    buf = nbuf;
  }
  dma_landed_barrier();                                // LDS is reused by the epilogue (and the tail DMAs of zeros are in)
  if constexpr (BLOCKED) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] += tot[i][j];
  }

  // ---- image layer folded into this layer's epilogue (ConvArgs::img_part)
  if constexpr (BREG && EPI == EPI_MUL && PREC == PREC_BF16X3 && TM == 2 && TN == 1 && WM == 2 && WN == 2) {
    if (a.img_part) {
      static_assert(2 * STAGE >= 128 * 64, "the C tile (then S_1 as pairs, in place; then T) inside the staging LDS");
      constexpr int TS2 = 57;                             // T row stride (54 used; odd: conflict-free column reads)
      float* Cs = smem;                                   // [128 rows][64] fp32 — rewritten IN PLACE as S_1's bf16 pairs:
      // row r keeps its 256 B = [chunk 0 | chunk 1] x 128 B, 16 B slots XOR-swizzled with (r >> 1) & 7 like an A tile.  The eight
      // lanes that own a row are neighbours in one wave and LDS serves a wave's instructions in order, so all of a row's
      // reads are done before any of its writes.
      const float inv_tw2 = 1.0f / (float)a.tw;
      // everything this epilogue needs from global memory is requested up front — the gates of this thread's four
      // (row, channel group) items and this wave's fragments of the tap matrix — so that the latencies overlap each other
      // and the LDS passes below instead of sitting between them
      f32x4 gq[4][2];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int item = tid + k * NT, row = item >> 3, g = item & 7;
        int ty, tx;
        divmod(row, a.tw, inv_tw2, ty, tx);
        const int Y = Y0 + ty, w = x0 + tx;
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        gq[k][0] = gq[k][1] = z4;                         // (gate = 0 for padding rows / pixels outside the image)
        if (row < a.th * a.tw && Y < yend && w < a.W) {
          int n, h;
          divmod(Y, a.H, inv_H, n, h);
          const int img = a.row2img ? a.row2img[n] : n;
          const float* gp = a.aux + ((size_t)img * HW + h * a.W + w) * a.N + g * 8;
          gq[k][0] = *reinterpret_cast<const f32x4*>(gp); gq[k][1] = *reinterpret_cast<const f32x4*>(gp + 4);
        }
      }
      u32x4 wq[2][2][2][2];                               // [chunk][step][hi|lo][n tile]
      {
        const int h2 = lane >> 5;
#pragma unroll
        for (int cc = 0; cc < 2; ++cc)
#pragma unroll
          for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int hl = 0; hl < 2; ++hl)
#pragma unroll
              for (int j = 0; j < 2; ++j)
                wq[cc][st][hl][j] = *reinterpret_cast<const u32x4*>(a.img_w + (size_t)(j * 32 + (lane & 31)) * 64 + cc * 32 + (4 * st + 2 * h2 + hl) * 4);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          Cs[((wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 64 + wn * 32 + (lane & 31)] = acc[i][0][r];
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 4; ++k) {                       // 128 rows x 8 channel groups = 1024 items over 256 threads
        const int item = tid + k * NT, row = item >> 3, g = item & 7;
        float* rp = Cs + row * 64 + (g >> 2) * 32;
        const f32x4 c0 = *reinterpret_cast<const f32x4*>(rp + (g & 3) * 8), c1 = *reinterpret_cast<const f32x4*>(rp + (g & 3) * 8 + 4);
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = c0[e] * gq[k][0][e]; v[4 + e] = c1[e] * gq[k][1][e]; }
        bf16x8 hi, lo;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          hi[q] = (__bf16)v[q];
          lo[q] = (__bf16)(v[q] - (float)hi[q]);
        }
        const int d = ((2 * (g & 3)) ^ ((row >> 1) & 7)) << 2;
        *reinterpret_cast<u32x4*>(rp + d) = __builtin_bit_cast(u32x4, hi);
        *reinterpret_cast<u32x4*>(rp + (d ^ 4)) = __builtin_bit_cast(u32x4, lo);
      }
      __syncthreads();
      // T = S_1 . W: wave w owns tile rows 32 w .. 32 w + 31, all 64 columns; K = 64 = 2 chunks x 2 steps
      f32x16 t2[2];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) t2[j][r] = 0.f;
      {
        const int a_row = wave * 32 + (lane & 31), swz2 = (lane >> 1) & 7, h2 = lane >> 5;
#pragma unroll
        for (int cc = 0; cc < 2; ++cc)
#pragma unroll
          for (int st = 0; st < 2; ++st) {
            const bf16x8 ah = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Cs + a_row * 64 + cc * 32 + (((4 * st + 2 * h2) ^ swz2) << 2)));
            const bf16x8 al = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Cs + a_row * 64 + cc * 32 + (((4 * st + 2 * h2 + 1) ^ swz2) << 2)));
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const bf16x8 bh = __builtin_bit_cast(bf16x8, wq[cc][st][0][j]), bl = __builtin_bit_cast(bf16x8, wq[cc][st][1][j]);
              t2[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, t2[j], 0, 0, 0);
              t2[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, t2[j], 0, 0, 0);
              t2[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, t2[j], 0, 0, 0);
            }
          }
      }
      __syncthreads();                                    // every wave is done reading A2: T may overwrite it
      float* Ts = smem;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int lr = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int col = j * 32 + (lane & 31);
          if (col < TS2) Ts[lr * TS2 + col] = t2[j][r];
        }
      }
      __syncthreads();
      // partial sums of the ring-inclusive positions: output (oy, ox) in [-1, th] x [-1, tw] tile coordinates gets
      // sum over taps of T[source = (oy - (kh - 1), ox - (kw - 1))][tap] for the sources this tile owns
      const int RW = a.tw + 2, npos = (a.th + 2) * RW;
      const int mtp_ = a.tile_map ? a.tile_map[mt] : mt;
      for (int p = tid; p < npos; p += NT) {
        const int ry = p / RW, rx = p - ry * RW, oy = ry - 1, ox = rx - 1;
        float pos[3] = {0.f, 0.f, 0.f}, neg[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const int sy = oy - (tap / 3 - 1), sx = ox - (tap % 3 - 1);
          if (sy >= 0 && sy < a.th && sx >= 0 && sx < a.tw && Y0 + sy < yend && x0 + sx < a.W) {
            const float* r = Ts + (sy * a.tw + sx) * TS2 + tap * 6;
            pos[0] += r[0]; pos[1] += r[1]; pos[2] += r[2];
            neg[0] += r[3]; neg[1] += r[4]; neg[2] += r[5];
          }
        }
        float* o = a.img_part + ((size_t)mtp_ * npos + p) * 6;
        o[0] = pos[0]; o[1] = pos[1]; o[2] = pos[2]; o[3] = neg[0]; o[4] = neg[1]; o[5] = neg[2];
      }
      return;
    }
  }

  // ---- conv-LRP epilogues: stage the C tile through the (now idle) LDS so that the gate loads
  // and the relevance stores are 16 B per lane along the channel axis (a pixel's channels are
  // contiguous in NHWC) instead of one dword per lane.
  if constexpr (EPI == EPI_IMG_STENCIL) {
    // row stride 65 floats: with 64 (= 256 B = one sweep of the banks) the stencil's reads of one column across the
    // 64 pixels of a wave all hit the same bank  [MI355X: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.75]
    constexpr int TS = BN + 1;
    static_assert(BM == IMG_PATCH * IMG_PATCH && BN == 64 && BM * TS <= 2 * STAGE, "patch tile is 256 x 64");
    float* Ts = smem;                                   // T of the patch: [256 patch pixels][64 columns (54 used)]
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int lr = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
#pragma unroll
        for (int j = 0; j < TN; ++j) Ts[lr * TS + (wn * TN + j) * 32 + (lane & 31)] = acc[i][j][r];
      }
    __syncthreads();
    // R_img[p] = sum_tap T[p - d(tap)][tap], d = (kh - 1, kw - 1): patch pixel (oy + 2 - kh, ox + 2 - kw)
    for (int o = tid; o < IMG_TILE * IMG_TILE; o += NT) {
      const int oy = o / IMG_TILE, ox = o - oy * IMG_TILE;
      const int y = py0 + oy, x = px0 + ox;
      if (y >= a.H || x >= a.W) continue;
      float pos[3] = {0.f, 0.f, 0.f}, neg[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const float* r = Ts + ((oy + 2 - tap / 3) * IMG_PATCH + (ox + 2 - tap % 3)) * TS + tap * 6;
        pos[0] += r[0]; pos[1] += r[1]; pos[2] += r[2];
        neg[0] += r[3]; neg[1] += r[4]; neg[2] += r[5];
      }
      const int img = a.row2img ? a.row2img[pn] : pn;
      if constexpr (PREC == PREC_F16X2) {               // undo the token's scale: the chain ends here
        const float sc = a.tok_fac[pn];
#pragma unroll
        for (int c = 0; c < 3; ++c) { pos[c] *= sc; neg[c] *= sc; }
      }
      const float* xv = a.ximg + ((size_t)img * HW + (size_t)y * a.W + x) * 3;
      float* ov = a.out + ((size_t)pn * HW + (size_t)y * a.W + x) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        if (a.img_mode == 0) ov[c] = xv[c] >= 0.f ? xv[c] * pos[c] : xv[c] * neg[c];   // LRP alpha1beta0 at the image
        else if (a.img_mode == 1) ov[c] = pos[c] + neg[c];                                // plain gradient
        else ov[c] = xv[c] * (pos[c] + neg[c]);                                           // input x gradient
      }
    }
    return;
  } else if constexpr (EPI != EPI_STORE) {
    // The C tile goes through the staging LDS in NH row slabs (a 256-row tile does not fit at once).
    constexpr int NH = (BM * BN + 2 * STAGE - 1) / (2 * STAGE);
    constexpr int RH = BM / NH;                         // rows per slab
    static_assert(BM % NH == 0 && RH % (TM * 32) == 0 && RH * BN <= 2 * STAGE, "slab must hold whole wave tiles");
    float* Cs = smem;
    constexpr bool SPLIT_OUT = SPLIT && (EPI == EPI_MUL || EPI == EPI_MUL_UP2);
    constexpr int CW = SPLIT_OUT ? 8 : 4;               // channels per thread (split8 output is written per group)
    constexpr int C4 = BN / CW, RPP = NT / C4;
    const int c4 = tid % C4, rin = tid / C4;
    const int col = n0 + c4 * CW;
    const int tn0 = m0 / HW, tp0 = m0 - tn0 * HW;
    const float invw = 1.0f / (float)a.W;
    // tile row lr -> linear NHWC row / image slot n / pixel (h, w); false = padding row
    auto locate = [&](int lr, int& row, int& n, int& h, int& w) -> bool {
      if constexpr (HALO) {
        int ty, tx;
        divmod(lr, a.tw, inv_tw, ty, tx);
        const int Y = Y0 + ty;
        w = x0 + tx;
        if (lr >= a.th * a.tw || Y >= yend || w >= a.W) return false;
        divmod(Y, a.H, inv_H, n, h);
        row = Y * a.W + w;
        return true;
      } else {
        row = m0 + lr;
        if (row >= a.M) return false;
        if constexpr (EPI == EPI_MUL || EPI == EPI_MUL_UP2) {
          n = tn0;
          int pix = tp0 + lr;
          while (pix >= HW) { pix -= HW; ++n; }
          h = (int)(((float)pix + 0.5f) * invw);
          w = pix - h * a.W;
          if (w < 0) { --h; w += a.W; } else if (w >= a.W) { ++h; w -= a.W; }
        }
        return true;
      }
    };
    // PREC_F16X2: largest |stored output| per token, gathered per block in LDS (slot = token - first token of the tile) and
    // flushed with ONE global atomic per token and block (a global atomic per wave and pass cost 0.4 ms on block1_conv2)
    // (the slots live in the staging LDS behind the C slab where the tile shape leaves room — every tile does except the
    // non-halo 128 x 128 / 256 x 256 ones, which fall back to the per-wave global atomic; LDS is sized to the byte:
    // two 128 x 128 halo blocks fill the CU's 160 KB exactly)
    constexpr bool TMAX_LDS = PREC == PREC_F16X2 && RH * BN + 16 <= 2 * STAGE;
    unsigned* tmax_s = reinterpret_cast<unsigned*>(smem + RH * BN);
    const int tok_first = HALO ? img0 : tn0;
    if constexpr (TMAX_LDS) {
      if (tid < 16) tmax_s[tid] = 0u;                   // (ordered before its first use by the barrier that publishes the C slab)
    }
#pragma unroll 1
    for (int hf = 0; hf < NH; ++hf) {
      if (hf) __syncthreads();                          // previous slab fully consumed
      if ((wm * TM * 32) / RH == hf) {
        // a wave's TM x 32 rows lie inside ONE slab (static_assert above), so their position in the slab does not depend on hf:
        // one base address + compile-time offsets (written with `- hf * RH` hipcc kept sixteen addresses per wave alive across
        // the main loop of the 8-wave tile, spilled them, and reloaded each behind its own s_waitcnt vmcnt(0))
        float* cw = Cs + (((wm * TM * 32) % RH) + 4 * (lane >> 5)) * BN + wn * TN * 32 + (lane & 31);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int j = 0; j < TN; ++j) cw[(i * 32 + (r & 3) + 8 * (r >> 2)) * BN + j * 32] = acc[i][j][r];
      }
      __syncthreads();
      if constexpr (EPI == EPI_BIAS || EPI == EPI_BIAS_RELU) {
        if (col < a.N) {
          const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
          const f32x4 bv = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + col) : zero4;
          float unscale = 1.f;                            // PREC_F16X2: the input's power-of-two scale, undone on the accumulator
          float omax = 0.f;
          if constexpr (PREC == PREC_F16X2) unscale = *a.in_unscale;
#pragma unroll 4
          for (int ps = 0; ps < RH / RPP; ++ps) {
            const int ll = rin + ps * RPP;
            int row, n_, h_, w_;
            if (!locate(hf * RH + ll, row, n_, h_, w_)) continue;
            f32x4 v = *reinterpret_cast<const f32x4*>(Cs + ll * BN + c4 * 4);
            if constexpr (PREC == PREC_F16X2) v *= unscale;
            v += bv;
            if (a.addend) v += *reinterpret_cast<const f32x4*>(a.addend + (size_t)row * a.N + col);   // second pass of a product
            if constexpr (EPI == EPI_BIAS) {
              if (a.gate_src) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(a.gate_src + (size_t)row * a.N + col);
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = av[q] / (v[q] + (v[q] == 0.f ? 1e-7f : 0.f));
              }
            }
            if constexpr (EPI == EPI_BIAS_RELU) {
#pragma unroll
              for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
            }
            if constexpr (PREC == PREC_F16X2) omax = fmaxf(omax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
            *reinterpret_cast<f32x4*>(a.out + (size_t)row * a.N + col) = v;
          }
          if constexpr (PREC == PREC_F16X2) {
            if (a.act_max_out) {                          // one atomic per wave, spread over ACT_MAX_SLOTS addresses
#pragma unroll
              for (int o = 32; o > 0; o >>= 1) omax = fmaxf(omax, __shfl_xor(omax, o));
              if (lane == 0 && omax > 0.f) atomicMax(a.act_max_out + ((blockIdx.x + wave) & (ACT_MAX_SLOTS - 1)), __float_as_uint(omax));
            }
          }
        }
      } else if constexpr (EPI == EPI_FWD_DUAL) {
        if (a.dual_il) {
          // interleaved columns: tile column p -> block p / 32; even block = c of channels ch.., the odd block behind it =
          // Z+ of the same channels.  The two threads that own a channel quad (one per block) share its rows by parity.
          const int p = c4 * 4, half = (p >> 5) & 1;
          const int ch = (n0 >> 1) + (p >> 6) * 32 + (p & 31);
          bool pooled = false;
          if constexpr (HALO && NH == 1 && PREC == PREC_F16X2) {
            if (a.pool_gc) {
              // ---- fused 2x2 max-pool (ConvArgs::pool_gc): item = (window of the tile, channel quad)
              pooled = true;
              const int wpr = a.tw >> 1, nq = BN / 8;     // windows per tile row; channel quads of the tile's c columns
              const int nitems = (a.th >> 1) * wpr * nq;
              const float inv_wpr = 1.0f / (float)wpr;
              float pmax0 = 0.f, pmax1 = 0.f;
              for (int it = tid; it < nitems; it += NT) {
                const int q = it % nq, win = it / nq;
                int wy, wx;
                divmod(win, wpr, inv_wpr, wy, wx);
                const int cq = 4 * q, chp = (n0 >> 1) + cq, pcp = (cq >> 5) * 64 + (cq & 31);
                const int r00 = (2 * wy) * a.tw + 2 * wx;
                int row = 0, n_ = 0, h_ = 0, w_ = 0;
                const bool ok = locate(r00, row, n_, h_, w_) && chp < a.split;
                const int rel = a.scale_per_img ? n_ - simg0 : 0;
                float unscale = rel == 0 ? us0 : us1, pscale = rel == 0 ? ps0 : ps1;
                if (ok && rel > 1) { unscale = a.in_unscale[simg0 + rel]; pscale = a.pairs_scale[simg0 + rel]; }
                f32x4 bvp = {0.f, 0.f, 0.f, 0.f};
                if (ok) bvp = *reinterpret_cast<const f32x4*>(a.bias + chp);
                f32x4 mx, zx;
                unsigned posw = 0u;
#pragma unroll
                for (int pp = 0; pp < 4; ++pp) {
                  const float* cr = Cs + (r00 + (pp >> 1) * a.tw + (pp & 1)) * BN + pcp;
                  f32x4 vc = *reinterpret_cast<const f32x4*>(cr), vz = *reinterpret_cast<const f32x4*>(cr + 32);
                  vc *= unscale; vz *= unscale;
                  vc += bvp; vz += bvp;
#pragma unroll
                  for (int e = 0; e < 4; ++e) {
                    const float av = a.dual_norelu ? vc[e] : fmaxf(vc[e], 0.f);
                    if (pp == 0 || av > mx[e]) {           // first maximum in scan order (what tf.gradients routes to)
                      mx[e] = av; zx[e] = vz[e];
                      posw = (posw & ~(0xFFu << (8 * e))) | ((unsigned)pp << (8 * e));
                    }
                  }
                }
                f32x4 gq;
#pragma unroll
                for (int e = 0; e < 4; ++e) gq[e] = mx[e] / (zx[e] + (zx[e] == 0.f ? 1e-7f : 0.f));
                if (ok) {
                  const size_t po = (((size_t)n_ * (a.H >> 1) + (h_ >> 1)) * (a.W >> 1) + (w_ >> 1)) * a.split + chp;
                  *reinterpret_cast<f32x4*>(a.pool_gc + po) = gq;
                  *reinterpret_cast<unsigned*>(a.pool_pos + po) = posw;
                  if (a.pool_x) *reinterpret_cast<f32x4*>(a.pool_x + po) = mx;
                  const float m4 = fmaxf(fmaxf(fabsf(mx[0]), fabsf(mx[1])), fmaxf(fabsf(mx[2]), fabsf(mx[3])));
                  if (rel == 0) pmax0 = fmaxf(pmax0, m4);
                  else if (rel == 1) pmax1 = fmaxf(pmax1, m4);
                  else if (m4 > 0.f && a.act_max_out) atomicMax(a.act_max_out + (size_t)(simg0 + rel) * ACT_MAX_SLOTS + ((blockIdx.x + wave) & (ACT_MAX_SLOTS - 1)), __float_as_uint(m4));
                }
                {
                  // pairs of the pooled activation: the two lanes of a split8 group (quads q, q ^ 1: neighbouring items of the
                  // same window) swap halves and store 16 B each, as in the unpooled epilogue below
                  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
                  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                  f16x4 hi, lo;
#pragma unroll
                  for (int e = 0; e < 4; ++e) {
                    const float sv = mx[e] * pscale;
                    hi[e] = (_Float16)sv;
                    lo[e] = (_Float16)(sv - (float)hi[e]);
                  }
                  const bool odd = q & 1;
                  const u32x2 mine = __builtin_bit_cast(u32x2, odd ? lo : hi);
                  const u32x2 give = __builtin_bit_cast(u32x2, odd ? hi : lo);
                  u32x2 got;
                  got[0] = __shfl_xor(give[0], 1);
                  got[1] = __shfl_xor(give[1], 1);
                  u32x4 v16;
                  v16[0] = odd ? got[0] : mine[0]; v16[1] = odd ? got[1] : mine[1];
                  v16[2] = odd ? mine[0] : got[0]; v16[3] = odd ? mine[1] : got[1];
                  if (ok) {
                    const size_t po = (((size_t)n_ * (a.H >> 1) + (h_ >> 1)) * (a.W >> 1) + (w_ >> 1)) * a.split + (chp & ~7);
                    *reinterpret_cast<u32x4*>(a.pairs_out + po + (odd ? 4 : 0)) = v16;
                  }
                }
              }
              if (a.act_max_out) {
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                  pmax0 = fmaxf(pmax0, __shfl_xor(pmax0, o));
                  pmax1 = fmaxf(pmax1, __shfl_xor(pmax1, o));
                }
                unsigned* ms = a.act_max_out + (size_t)simg0 * ACT_MAX_SLOTS + ((blockIdx.x + wave) & (ACT_MAX_SLOTS - 1));
                if (lane == 0 && pmax0 > 0.f) atomicMax(ms, __float_as_uint(pmax0));
                if (lane == 0 && pmax1 > 0.f) atomicMax(ms + ACT_MAX_SLOTS, __float_as_uint(pmax1));
              }
            }
          }
          if (!pooled && ch < a.split) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + ch);
            const int pc = (p & ~63) + (p & 31);          // the c column of the pair; Z+ sits 32 further
            // running max|a_l| of this thread's rows of images simg0 .. simg0 + 3 (a tile spans more than two images where an
            // image is a few dozen rows — the 7 x 7 maps of a ResNet's last stack: as unreduced atomics per row those launches
            // took 250 us instead of 36 [MI355X])
            float omax0 = 0.f, omax1 = 0.f, omax2 = 0.f, omax3 = 0.f;
            auto max_slot = [&](int img) { return a.act_max_out + (size_t)img * ACT_MAX_SLOTS + ((blockIdx.x + wave) & (ACT_MAX_SLOTS - 1)); };
            static_assert((RH / RPP) % 4 == 0, "constant, even trip count: the loop holds lane shuffles (no remainder loop)");
#pragma unroll 2
            for (int pi = 0; pi < RH / RPP / 2; ++pi) {
              const int ll = rin + (half + 2 * pi) * RPP;
              int row, n_, h_, w_;
              if (!locate(hf * RH + ll, row, n_, h_, w_)) continue;
              int rimg = 0;                                // image of the row (1-tap launches over a pixel list: rows / image is
              if (a.scale_per_img) {                       //  not a power of two — float estimate + fix-up, not an integer division)
                if constexpr (HALO) rimg = n_;
                else { int rr; divmod(row, a.img_rows, inv_img_rows, rimg, rr); }
              }
              const int rel = rimg - simg0;
              float unscale = rel == 0 ? us0 : us1, pscale = rel == 0 ? ps0 : ps1;
              if (__builtin_expect(rel > 1, 0)) {         // (a tile over three or more images: images smaller than half a tile)
                if constexpr (PREC == PREC_F16X2) unscale = a.in_unscale[simg0 + rel];
                if (a.pairs_out) pscale = a.pairs_scale[simg0 + rel];
              }
              f32x4 vc = *reinterpret_cast<const f32x4*>(Cs + ll * BN + pc);
              f32x4 vz = *reinterpret_cast<const f32x4*>(Cs + ll * BN + pc + 32);
              if constexpr (PREC == PREC_F16X2) { vc *= unscale; vz *= unscale; }
              vc += bv; vz += bv;
              if (!a.dual_norelu) {
#pragma unroll
                for (int q = 0; q < 4; ++q) vc[q] = fmaxf(vc[q], 0.f);
              }
              if (a.dual_gate == 2) {
                const f32x4 ga = *reinterpret_cast<const f32x4*>(a.bn_gamma + ch), be = *reinterpret_cast<const f32x4*>(a.bn_beta + ch);
                const f32x4 mu = *reinterpret_cast<const f32x4*>(a.bn_mean + ch), va = *reinterpret_cast<const f32x4*>(a.bn_var + ch);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  const float cv = vc[q];
                  const float y = ga[q] * (cv - mu[q]) / sqrtf(va[q] + a.bn_eps) + be[q];
                  const float d = (cv - mu[q]) * y;
                  const float ds = d + (d >= 0.f ? 1e-7f : -1e-7f);                      // sign stabiliser, then SafeDivide
                  const float qq = (cv * (y - be[q])) / (ds + (ds == 0.f ? 1e-7f : 0.f)) / (vz[q] + (vz[q] == 0.f ? 1e-7f : 0.f));
                  const float av = a.bn_relu ? fmaxf(y, 0.f) : y;
                  vc[q] = av;
                  vz[q] = a.bn_relu ? av * qq : qq;
                }
              } else if (a.dual_gate) {
#pragma unroll
                for (int q = 0; q < 4; ++q) vz[q] = vc[q] / (vz[q] + (vz[q] == 0.f ? 1e-7f : 0.f));
              }
              if (a.act_max_out) {
                const float m4 = fmaxf(fmaxf(fabsf(vc[0]), fabsf(vc[1])), fmaxf(fabsf(vc[2]), fabsf(vc[3])));
                if (rel == 0) omax0 = fmaxf(omax0, m4);
                else if (rel == 1) omax1 = fmaxf(omax1, m4);
                else if (rel == 2) omax2 = fmaxf(omax2, m4);
                else if (rel == 3) omax3 = fmaxf(omax3, m4);
                else if (m4 > 0.f) atomicMax(max_slot(simg0 + rel), __float_as_uint(m4));
              }
              if (!a.skip_out) *reinterpret_cast<f32x4*>(a.out + (size_t)row * a.split + ch) = vc;
              *reinterpret_cast<f32x4*>(a.out2 + (size_t)row * a.split + ch) = vz;
              {
                if (a.pairs_out) {
                  // This thread's 4 channels are one half of a split8 group [8 x fp16 hi | 8 x fp16 lo]; the other half sits
                  // in the neighbouring lane (same row, next channel quad: c4 ^ 1).  The two swap halves so that each issues
                  // ONE 16-byte store (the even lane the group's hi half, the odd lane its lo half) — as 8-byte stores the
                  // epilogue was store-issue-bound [MI355X: the forward convs 8-10 % slower].
                  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
                  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                  f16x4 hi, lo;
#pragma unroll
                  for (int q = 0; q < 4; ++q) {
                    const float sv = vc[q] * pscale;
                    hi[q] = (_Float16)sv;
                    lo[q] = (_Float16)(sv - (float)hi[q]);
                  }
                  const bool odd = (ch >> 2) & 1;
                  const u32x2 mine = __builtin_bit_cast(u32x2, odd ? lo : hi);      // what this lane keeps
                  const u32x2 give = __builtin_bit_cast(u32x2, odd ? hi : lo);      // what the neighbour stores
                  u32x2 got;
                  got[0] = __shfl_xor(give[0], 1);
                  got[1] = __shfl_xor(give[1], 1);
                  u32x4 v16;
                  v16[0] = odd ? got[0] : mine[0]; v16[1] = odd ? got[1] : mine[1];     // channels 0-3 of the group first
                  v16[2] = odd ? mine[0] : got[0]; v16[3] = odd ? mine[1] : got[1];
                  float* grp = a.pairs_out + (size_t)row * a.split + (ch & ~7);
                  *reinterpret_cast<u32x4*>(grp + (odd ? 4 : 0)) = v16;
                }
              }
            }
            if (a.act_max_out) {                            // (any PREC: the image layer's fp32 GEMM raises its maximum here too)
#pragma unroll
              for (int o = 32; o > 0; o >>= 1) {
                omax0 = fmaxf(omax0, __shfl_xor(omax0, o));
                omax1 = fmaxf(omax1, __shfl_xor(omax1, o));
                omax2 = fmaxf(omax2, __shfl_xor(omax2, o));
                omax3 = fmaxf(omax3, __shfl_xor(omax3, o));
              }
              if (lane == 0 && omax0 > 0.f) atomicMax(max_slot(simg0), __float_as_uint(omax0));
              if (lane == 0 && omax1 > 0.f) atomicMax(max_slot(simg0 + 1), __float_as_uint(omax1));
              if (lane == 0 && omax2 > 0.f) atomicMax(max_slot(simg0 + 2), __float_as_uint(omax2));
              if (lane == 0 && omax3 > 0.f) atomicMax(max_slot(simg0 + 3), __float_as_uint(omax3));
            }
          }
        } else
        // cols [0,split) -> out = relu(acc + bias)  (a_l);  cols [split,2 split) -> out2 = acc + bias  (Z+_l)
        if (col < 2 * a.split) {
          const bool isz = col >= a.split;
          const int c = isz ? col - a.split : col;
          const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + c);
          float* dst = (isz ? a.out2 : a.out) + c;
          float unscale = 1.f, omax = 0.f;                // PREC_F16X2: as in the BIAS epilogues (max of the a half only)
          if constexpr (PREC == PREC_F16X2) unscale = *a.in_unscale;
#pragma unroll 4
          for (int ps = 0; ps < RH / RPP; ++ps) {
            const int ll = rin + ps * RPP;
            int row, n_, h_, w_;
            if (!locate(hf * RH + ll, row, n_, h_, w_)) continue;
            f32x4 v = *reinterpret_cast<const f32x4*>(Cs + ll * BN + c4 * 4);
            if constexpr (PREC == PREC_F16X2) v *= unscale;
            v += bv;
            if (!isz && !a.dual_norelu) {
#pragma unroll
              for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
            }
            if constexpr (PREC == PREC_F16X2) {
              if (!isz) omax = fmaxf(omax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
            }
            *reinterpret_cast<f32x4*>(dst + (size_t)row * a.split) = v;
          }
          if constexpr (PREC == PREC_F16X2) {
            if (a.act_max_out) {
#pragma unroll
              for (int o = 32; o > 0; o >>= 1) omax = fmaxf(omax, __shfl_xor(omax, o));
              if (lane == 0 && omax > 0.f) atomicMax(a.act_max_out + ((blockIdx.x + wave) & (ACT_MAX_SLOTS - 1)), __float_as_uint(omax));
            }
          }
        }
      } else {
        // out = acc * gate (fp32), stored as fp32 or re-split into [hi8 | lo8] for the next layer's MFMAs.
        // The gate values are two dependent global loads away (row2img[n], then the gate row) and nothing in a pass
        // depends on the pass before it, so the loads are issued in bulk: every pass's image slot first, then the
        // gate rows one pass AHEAD of the arithmetic (written as explicit phases — left as load/multiply/store per
        // pass, the possible aliasing of `out` and `aux` made hipcc serialise a full memory round trip per pass, which
        // the short K = 576 layers could not hide).
        if (col < a.N) {
          // ---- the walk's common case (plain gate, no residual join, no second head, no gradient-walk switches): the same
          // arithmetic with nothing to decide inside the passes.  In the general form below every runtime switch sits next to
          // a load, hipcc branches around those loads, cannot count them any more and drains the queue (s_waitcnt vmcnt(0) /
          // (1)) in every pass — the gate rows requested "one pass ahead" were in fact waited for immediately, and an 8-wave
          // tile's epilogue ran as 16 dependent L2 round trips with the matrix pipe idle.  Here the gate loads of RING passes
          // are in flight, unconditionally (padding rows read row 0 of image 0), and each pass waits with a counted vmcnt.
          // Rows are located twice (at issue and at use: a few VALU ops) instead of keeping six arrays of NP entries alive.
          if constexpr (PREC != PREC_F16X2) {
            const bool tail_ = EPI == EPI_MUL && a.join != nullptr, head2_ = EPI == EPI_MUL && a.out2s != nullptr;
            if (!(EPI == EPI_MUL && a.gate_none) && !tail_ && !head2_ && !a.gate_binary && !a.relu_out && !a.epi_generic) {
              constexpr int NPf = RH / RPP;
              constexpr int UPf = EPI == EPI_MUL_UP2 ? 4 : 1;
              constexpr int RING = (UPf == 1 ? 4 : 2) < NPf ? (UPf == 1 ? 4 : 2) : NPf;
              const int W2f = 2 * a.W, H2f = 2 * a.H;
              int imgf[NPf];
#pragma unroll
              for (int ps = 0; ps < NPf; ++ps) {
                int row, n_, h_, w_;
                if (!locate(hf * RH + rin + ps * RPP, row, n_, h_, w_)) n_ = 0;
                imgf[ps] = a.row2img ? a.row2img[n_] : n_;
              }
              f32x4 gq[RING][UPf][CW / 4];
              auto issue = [&](int ps) {
                int row, n_, h_, w_;
                if (!locate(hf * RH + rin + ps * RPP, row, n_, h_, w_)) { h_ = 0; w_ = 0; }
#pragma unroll
                for (int q = 0; q < UPf; ++q) {
                  const float* gp = EPI == EPI_MUL ? a.aux + ((size_t)imgf[ps] * HW + h_ * a.W + w_) * a.N + col
                                                   : a.aux + (((size_t)imgf[ps] * H2f + 2 * h_ + (q >> 1)) * W2f + 2 * w_ + (q & 1)) * a.N + col;
#pragma unroll
                  for (int q4 = 0; q4 < CW / 4; ++q4) gq[ps % RING][q][q4] = *reinterpret_cast<const f32x4*>(gp + 4 * q4);
                }
              };
#pragma unroll
              for (int ps = 0; ps < RING - 1; ++ps) issue(ps);
              const bool plain = SPLIT_OUT && a.out_plain;
#pragma unroll
              for (int ps = 0; ps < NPf; ++ps) {
                if (ps + RING - 1 < NPf) issue(ps + RING - 1);
                int row, n_, h_, w_;
                if (locate(hf * RH + rin + ps * RPP, row, n_, h_, w_)) {
                  const int ll = rin + ps * RPP;
                  float v[CW];
#pragma unroll
                  for (int q4 = 0; q4 < CW / 4; ++q4)
                    *reinterpret_cast<f32x4*>(v + 4 * q4) = *reinterpret_cast<const f32x4*>(Cs + ll * BN + c4 * CW + 4 * q4);
#pragma unroll
                  for (int q = 0; q < UPf; ++q) {
                    float r[CW];
#pragma unroll
                    for (int q4 = 0; q4 < CW / 4; ++q4)
#pragma unroll
                      for (int e = 0; e < 4; ++e) r[4 * q4 + e] = v[4 * q4 + e] * gq[ps % RING][q][q4][e];
                    float* dst = EPI == EPI_MUL
                                     ? a.out + (size_t)row * a.N + col
                                     : a.out + (((size_t)n_ * H2f + 2 * h_ + (q >> 1)) * W2f + 2 * w_ + (q & 1)) * a.N + col;
                    if constexpr (SPLIT_OUT) {
                      if (plain) {
                        *reinterpret_cast<f32x4*>(dst) = *reinterpret_cast<const f32x4*>(r);
                        *reinterpret_cast<f32x4*>(dst + 4) = *reinterpret_cast<const f32x4*>(r + 4);
                      } else {
                        split8_store(r, dst);
                      }
                    } else {
                      *reinterpret_cast<f32x4*>(dst) = *reinterpret_cast<const f32x4*>(r);
                    }
                  }
                }
              }
              continue;                                   // next slab
            }
            // (A counted-prefetch form of the join + second-head epilogue — the tail of a ResNet identity block — was built here in
            //  round 4: config 4 -0.5 ms, but its mere presence in this instantiation cost the VGG walk's launches 2.4 %
            //  [MI355X, same box: 24.55 -> 25.15 ms, WRITE_SIZE +11 %: the register allocator spilled], so it was taken out
            //  again; those launches are bound by their three full-width streams either way)
          }
          constexpr int NP = RH / RPP;
          constexpr int UPN = EPI == EPI_MUL_UP2 ? 4 : 1;
          const int W2 = 2 * a.W, H2 = 2 * a.H;
          int rowv[NP], nv[NP], hv[NP], wv[NP], imgv[NP];
          bool okv[NP];
#pragma unroll
          for (int ps = 0; ps < NP; ++ps) {
            okv[ps] = locate(hf * RH + rin + ps * RPP, rowv[ps], nv[ps], hv[ps], wv[ps]);
            if (!okv[ps]) { rowv[ps] = 0; nv[ps] = 0; hv[ps] = 0; wv[ps] = 0; }
            imgv[ps] = a.row2img ? a.row2img[nv[ps]] : nv[ps];
          }
          // PREC_F16X2: running maximum of |stored output| for the token of the rows this thread has seen so far
          unsigned lmax = 0u;
          int ltok = -1;
          auto flush_max = [&](int t, unsigned m) {
            if (m) {
              const int rel = t - tok_first;
              if (TMAX_LDS && rel >= 0 && rel < 16) atomicMax(tmax_s + rel, m);
              else atomicMax(a.tok_max_out + t, m);
            }
          };
          const bool tail = EPI == EPI_MUL && a.join != nullptr, head2 = EPI == EPI_MUL && a.out2s != nullptr;
          struct Gates { f32x4 g[UPN][CW / 4]; f32x4 jn[CW / 4], jg[CW / 4], g2[CW / 4]; };
          auto gate_ptr = [&](int ps, int q) -> const float* {
            if constexpr (EPI == EPI_MUL)
              return a.aux + ((size_t)imgv[ps] * HW + hv[ps] * a.W + wv[ps]) * a.N + col;
            else
              return a.aux + (((size_t)imgv[ps] * H2 + 2 * hv[ps] + (q >> 1)) * W2 + 2 * wv[ps] + (q & 1)) * a.N + col;
          };
          auto load_gates = [&](Gates& G, int ps) {
            const f32x4 one4 = {1.f, 1.f, 1.f, 1.f};
#pragma unroll
            for (int q = 0; q < UPN; ++q)
#pragma unroll
              for (int q4 = 0; q4 < CW / 4; ++q4)
                G.g[q][q4] = (EPI == EPI_MUL && a.gate_none) ? one4 : *reinterpret_cast<const f32x4*>(gate_ptr(ps, q) + 4 * q4);
            if constexpr (EPI == EPI_MUL) {
              const size_t go = ((size_t)imgv[ps] * HW + hv[ps] * a.W + wv[ps]) * a.N + col;
#pragma unroll
              for (int q4 = 0; q4 < CW / 4; ++q4) {
                if (tail) {
                  G.jn[q4] = *reinterpret_cast<const f32x4*>(a.join + (size_t)rowv[ps] * a.N + col + 4 * q4);
                  G.jg[q4] = *reinterpret_cast<const f32x4*>(a.join_gate + go + 4 * q4);
                }
                if (head2) G.g2[q4] = *reinterpret_cast<const f32x4*>(a.gate2 + go + 4 * q4);
              }
            }
          };
          Gates cur, nxt;
          load_gates(cur, 0);
#pragma unroll
          for (int ps = 0; ps < NP; ++ps) {
            if (ps + 1 < NP) load_gates(nxt, ps + 1);
            float fac16 = 1.f;                            // PREC_F16X2: the power of two this launch adds to the token's scale
            if constexpr (PREC == PREC_F16X2) {
              fac16 = a.tok_fac[nv[ps]];
              if (okv[ps] && nv[ps] != ltok) { flush_max(ltok, lmax); ltok = nv[ps]; lmax = 0u; }
            }
            if (okv[ps]) {
              const int ll = rin + ps * RPP;
              float v[CW];
#pragma unroll
              for (int q4 = 0; q4 < CW / 4; ++q4)
                *reinterpret_cast<f32x4*>(v + 4 * q4) = *reinterpret_cast<const f32x4*>(Cs + ll * BN + c4 * CW + 4 * q4);
#pragma unroll
              for (int q = 0; q < UPN; ++q) {
                float r[CW];
#pragma unroll
                for (int q4 = 0; q4 < CW / 4; ++q4) {
                  f32x4 g = cur.g[q][q4];
                  if (a.gate_binary) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) g[e] = g[e] != 0.f ? 1.f : 0.f;
                  }
#pragma unroll
                  for (int e = 0; e < 4; ++e) {
                    float pr = v[4 * q4 + e] * g[e];
                    if (tail) pr += cur.jn[q4][e] * cur.jg[q4][e];
                    if constexpr (PREC == PREC_F16X2) {
                      pr *= fac16;
                      lmax = max(lmax, __float_as_uint(fabsf(pr)));               // (non-negative floats order like their bits)
                    }
                    r[4 * q4 + e] = a.relu_out ? fmaxf(pr, 0.f) : pr;
                  }
                }
                float* dst = EPI == EPI_MUL
                                 ? a.out + (size_t)rowv[ps] * a.N + col
                                 : a.out + (((size_t)nv[ps] * H2 + 2 * hv[ps] + (q >> 1)) * W2 + 2 * wv[ps] + (q & 1)) * a.N + col;
                if constexpr (SPLIT_OUT) {
                  if (a.out_plain) {
                    *reinterpret_cast<f32x4*>(dst) = *reinterpret_cast<const f32x4*>(r);
                    *reinterpret_cast<f32x4*>(dst + 4) = *reinterpret_cast<const f32x4*>(r + 4);
                  } else if constexpr (PREC == PREC_F16X2) {
                    split8h_store(r, dst);
                  } else {
                    split8_store(r, dst);
                  }
                } else {
                  *reinterpret_cast<f32x4*>(dst) = *reinterpret_cast<const f32x4*>(r);
                }
                if (head2) {
                  float r2[CW];
#pragma unroll
                  for (int q4 = 0; q4 < CW / 4; ++q4)
#pragma unroll
                    for (int e = 0; e < 4; ++e) r2[4 * q4 + e] = r[4 * q4 + e] * cur.g2[q4][e];
                  float* d2 = a.out2s + (size_t)rowv[ps] * a.N + col;
                  if constexpr (SPLIT_OUT) {
                    split8_store(r2, d2);
                  } else {
                    *reinterpret_cast<f32x4*>(d2) = *reinterpret_cast<const f32x4*>(r2);
                  }
                }
              }
            }
            cur = nxt;
          }
          if constexpr (PREC == PREC_F16X2) flush_max(ltok, lmax);
        }
      }
    }
    if constexpr (TMAX_LDS && (EPI == EPI_MUL || EPI == EPI_MUL_UP2)) {
      __syncthreads();
      if (tid < 16 && tmax_s[tid]) atomicMax(a.tok_max_out + tok_first + tid, tmax_s[tid]);
    }
    return;
  }

  // ---- scalar epilogue (EPI_STORE, N = 54 is not a multiple of 4).  C/D map of the 32x32 MFMA:
  // col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const int col_base = n0 + wn * TN * 32 + (lane & 31);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const bool rv = row < a.M;
      if constexpr (EPI == EPI_STORE) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int col = col_base + j * 32;
          if (rv && col < a.N) a.out[(size_t)row * a.N + col] = acc[i][j][r];
        }
      }
    }
  }
#endif
}

#ifndef LRP_CONV_KERNEL_ONLY   // (a translation unit that wants ONE instantiation for ISA inspection defines this and skips the launchers)
// tile configurations: (WM,WN,TM,TN) -> BM x BN
//   big   : 2,2,2,2 -> 128 x 128   (N >= 128)                 64 KB LDS, 2 blocks/CU   [fp32]
//   n64   : 2,2,2,1 -> 128 x  64   (N == 64 layers)           48 KB LDS, 3 blocks/CU
//   n32   : 4,1,1,1 -> 128 x  32   (N <= 32: tiny test nets)  40 KB LDS
//   w256  : 2,4,4,2 -> 256 x 256   8 waves, 128 KB LDS, 1 block/CU   [bf16x3, N % 256 == 0]
//   w128  : 4,2,2,2 -> 256 x 128   8 waves,  96 KB LDS, 1 block/CU   [bf16x3, N % 128 == 0]
// bf16x3 spends 5.3x fewer matrix cycles per byte staged, so its limiter is the L2 -> LDS path
// (~43 GB/s per CU at 128 x 128): the 8-wave tiles stage 25 % / 50 % fewer bytes per FLOP.
struct ConvTile { int BM, BN; };
inline ConvTile conv_pick_tile(int N) {
  if (N > 64) return {128, 128};
  if (N > 32) return {128, 64};
  return {128, 32};
}
inline int conv_npad(int N) { ConvTile t = conv_pick_tile(N); return (N + t.BN - 1) / t.BN * t.BN; }
inline int conv_cinp(int Cin) { return (Cin + 31) / 32 * 32; }

// switch LRP_CONV_TILE: 0 = auto, 1 = never use the 8-wave tiles, 128 = cap them at 256 x 128
inline int conv_tile_override() { return sw().conv_tile; }

// small grids take 64 x 64 tiles (conv_launch_epi): LRP_CONV_SMALL=0 disables; they are used up to a 128-row grid of 128 tiles
// (4x as many 64 x 64 workgroups = the 512 that are resident at once; beyond that a second round of small tiles costs more
// than the large tiles' idle CUs [MI355X, one image, 252 large tiles: 115 -> 145 us])
constexpr int CONV_SMALL_NS = 4;                       // LDS stages of the 64 x 64 tile's k pipeline (4 x 16 KB: two workgroups per CU... see NS)
inline bool conv_small_tile_on() { return sw().conv_small != 0; }
constexpr long conv_small_tile_blocks() { return 128L; }

// halo-resident variant: switch LRP_CONV_HALO = 0 never, 1 (default) when a tile shape fills >= 90 % of the M tile,
// 2 always (tests: ragged tile shapes)
inline int conv_halo_mode() { return sw().conv_halo; }
// best (tw, th, pitch) for a BM-row tile on an H x W image stack; returns the fraction of MFMA rows doing real work
inline float conv_halo_geom(int BM, int H, int W, int& tw, int& th, int& hrows) {
  const int avail = conv_halo_rows(BM) / HALO_PITCH;   // resident image rows that fit
  float best = 0.f;
  for (int c = 1; c <= HALO_PITCH - 2 && c <= W; ++c) {
    // th stack rows cross at most ceil((th-1)/H) image boundaries, each costing one separator row
    int t = BM / c;
    while (t > 0 && t + 2 + (t - 1 + H - 1) / H > avail) --t;
    if (t < 1) continue;
    const int cols = (W + c - 1) / c;
    const float u = (float)(t * c) / (float)BM * (float)W / (float)(cols * c);
    if (u > best + 1e-6f) { best = u; tw = c; th = t; hrows = t + 2 + (t - 1 + H - 1) / H; }
  }
  return best;
}

// ... with an even number of rows and columns (ConvArgs::pool_gc: 2x2 windows never straddle two tiles)
inline float conv_halo_geom_even(int BM, int H, int W, int& tw, int& th, int& hrows) {
  const int avail = conv_halo_rows(BM) / HALO_PITCH;
  float best = 0.f;
  for (int c = 2; c <= HALO_PITCH - 2 && c <= W; c += 2) {
    int t = (BM / c) & ~1;
    while (t > 0 && t + 2 + (t - 1 + H - 1) / H > avail) t -= 2;
    if (t < 2) continue;
    const int cols = (W + c - 1) / c;
    const float u = (float)(t * c) / (float)BM * (float)W / (float)(cols * c);
    if (u > best + 1e-6f) { best = u; tw = c; th = t; hrows = t + 2 + (t - 1 + H - 1) / H; }
  }
  return best;
}

// Would a 3x3 MUL launch with N = n_out, input H x W, take the weights-in-registers kernel?  (Encoder::explain asks before it
// chooses the compact pool interface, which only that kernel reads.)
inline bool conv_takes_breg(int n_out, int H, int W, bool have_frag) {
  if (!sw().conv_breg || !have_frag || conv_halo_mode() <= 0 || n_out > 64 || n_out <= 32) return false;
  int tw, th, hrows;
  return conv_halo_geom(128, H, W, tw, th, hrows) >= 0.9f;
}

// The 8-wave tile width of a split-format reverse-walk launch: 256 (256 x 256 tile), 128 (256 x 128, LRP_CONV_TILE=128 only) or 0.
// ONE rule for conv_launch_epi and conv_takes_pw.
inline int conv_wide_tile(int n_out, long mrows) {
  if (n_out < 128 || (n_out % 128) != 0) return 0;
  int wide = (n_out % 256) == 0 ? 256 : 0;               // measured: 256 x 128 loses to two 128 x 128 blocks per CU
  if (conv_tile_override() == 128) wide = 128;
  if (conv_tile_override() == 1) wide = 0;
  // one 8-wave block per CU: only worth it when the grid still fills the chip ~1.5 times over
  if (wide && ((mrows + 255) / 256) * (n_out / wide) < 400) wide = 0;
  return wide;
}

// Can the interleaved dual forward of a layer with `cout` channels on NB images of H x W pool in its epilogue
// (ConvArgs::pool_gc)?  The launch must reach the 128 x 128 resident-image kernel: mirrors conv_launch_epi; LRP_POOL_FUSED=0 disables.
inline bool conv_takes_pool_fused(int cout, int NB, int H, int W) {
  if (!sw().pool_fused || conv_halo_mode() <= 0 || (H & 1) || (W & 1) || (cout & 31) || conv_pick_tile(2 * cout).BN != 128) return false;
  const long blocks = (((long)NB * H * W + 127) / 128) * ((2 * cout + 127) / 128);
  if (conv_small_tile_on() && blocks <= 2 * conv_small_tile_blocks()) return false;      // (small / mid-size grids: other tiles)
  int tw, th, hrows;
  return conv_halo_geom_even(128, H, W, tw, th, hrows) >= 0.8f;
}

// Would a split-bf16 3x3 MUL launch (N = n_out columns, NB x H x W rows) take a pipelined halo kernel — 128 x 128 or the 8-wave
// 256 x 256 — and fit its loader of the compact pool interface (two items per thread and channel chunk)?  Mirrors the tile
// choice of conv_launch_epi; LRP_UP2_PW=0 disables.
inline bool conv_takes_pw(int n_out, int NB, int H, int W) {
  if (!sw().up2_pw || conv_halo_mode() <= 0 || conv_pick_tile(n_out).BN != 128 || (H & 1) || (W & 1)) return false;
  const long mrows = (long)NB * H * W;
  int BM = 128, threads = 256;
  const int wide = conv_wide_tile(n_out, mrows);
  if (wide == 128) return false;                          // (LRP_CONV_TILE=128: the 256 x 128 tile has no resident-image variant)
  if (wide == 256) { BM = 256; threads = 512; }
  if (BM == 128 && conv_small_tile_on() && ((mrows + 127) / 128) * ((n_out + 127) / 128) <= conv_small_tile_blocks()) return false;
  int tw, th, hrows;
  if (conv_halo_geom(BM, H, W, tw, th, hrows) < 0.9f) return false;
  return hrows * ((tw + 2) / 2 + 1) * 4 <= 2 * threads;
}

// device table of a.order for this call's token -> image map (see TileOrder); nullptr when the stack order is as good
inline const int* conv_tile_order(const ConvArgs& a, hipStream_t st) {
  if (!sw().tile_order || !a.order || !a.row2img_host || a.NB < 2 || a.th < 1) return nullptr;
  TileOrder& o = *a.order;
  const int tiles_y = a.tpt > 0 ? a.NB * a.tpt : (a.nyh + a.th - 1) / a.th;
  const bool same = o.H == a.H && o.th == a.th && o.nyh == a.nyh && o.cols_t == a.cols_t && o.tpt == a.tpt && (int)o.sig.size() == a.NB &&
                    memcmp(o.sig.data(), a.row2img_host, (size_t)a.NB * sizeof(int)) == 0;
  if (same) return o.identity ? nullptr : o.dev;
  o.sig.assign(a.row2img_host, a.row2img_host + a.NB);
  o.H = a.H; o.th = a.th; o.nyh = a.nyh; o.cols_t = a.cols_t; o.tpt = a.tpt;
  constexpr int band = 1, strip = 2;                   // row band of `band` tile rows per token; column tiles per strip (measured, round 2)
  std::vector<long long> key((size_t)tiles_y);
  for (int ty = 0; ty < tiles_y; ++ty) {
    int Ym = a.tpt > 0 ? (ty / a.tpt) * a.H + (ty % a.tpt) * a.th + a.th / 2 : ty * a.th + a.th / 2;
    if (a.tpt > 0 && Ym > (ty / a.tpt + 1) * a.H - 1) Ym = (ty / a.tpt + 1) * a.H - 1;
    if (Ym > a.nyh - 1) Ym = a.nyh - 1;
    const int t = Ym / a.H, hpos = Ym - t * a.H;
    key[ty] = ((long long)o.sig[t] << 42) | ((long long)(hpos / (a.th * band)) << 21) | (long long)t;
  }
  std::vector<int> rows((size_t)tiles_y);
  for (int i = 0; i < tiles_y; ++i) rows[i] = i;
  std::stable_sort(rows.begin(), rows.end(), [&](int x, int y) { return key[x] < key[y]; });
  // rows are now (image, band, token); within an image, walk the column strips outermost
  const int sw = strip > 0 && strip < a.cols_t ? strip : a.cols_t;
  const size_t m_tiles = (size_t)tiles_y * a.cols_t;
  std::vector<int> map;
  map.reserve(m_tiles);
  for (int r0 = 0; r0 < tiles_y;) {
    int r1 = r0;
    while (r1 < tiles_y && (key[rows[r1]] >> 42) == (key[rows[r0]] >> 42)) ++r1;
    for (int c0 = 0; c0 < a.cols_t; c0 += sw)
      for (int r = r0; r < r1; ++r)
        for (int c = c0; c < c0 + sw && c < a.cols_t; ++c) map.push_back(rows[r] * a.cols_t + c);
    r0 = r1;
  }
  o.identity = true;
  for (size_t i = 0; i < m_tiles; ++i)
    if (map[i] != (int)i) { o.identity = false; break; }
  if (o.identity) return nullptr;
  if (o.ev) (void)hipEventSynchronize(o.ev);            // the staging buffer may still feed the previous upload
  if (o.cap < m_tiles) {
    if (o.dev) (void)hipFree(o.dev);
    if (o.pinned) (void)hipHostFree(o.pinned);
    o.dev = o.pinned = nullptr;
    o.cap = 0;
    if (hipMalloc((void**)&o.dev, m_tiles * sizeof(int)) != hipSuccess ||
        hipHostMalloc((void**)&o.pinned, m_tiles * sizeof(int), hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      o.sig.clear();                                    // (try again next call; this one runs in stack order)
      return nullptr;
    }
    o.cap = m_tiles;
  }
  memcpy(o.pinned, map.data(), m_tiles * sizeof(int));
  if (!o.ev) (void)hipEventCreateWithFlags(&o.ev, hipEventDisableTiming);
  if (hipMemcpyAsync(o.dev, o.pinned, m_tiles * sizeof(int), hipMemcpyHostToDevice, st) != hipSuccess) {
    (void)hipGetLastError();
    o.sig.clear();
    return nullptr;
  }
  (void)hipEventRecord(o.ev, st);
  return o.dev;
}

// EPI_IMG_STENCIL: NB = image slots (tokens), H x W = the image; in = S_1 (NB, H, W, Cin); N = 54 <= 64
template <int PREC>
inline hipError_t conv_launch_img(ConvArgs a, hipStream_t st) {
  if (a.taps != 1 || a.N > 64 || !a.ximg || (a.Cin & (PREC != PREC_FP32 ? 7 : 3))) return hipErrorInvalidValue;
  if (PREC == PREC_F16X2 && !a.tok_fac) return hipErrorInvalidValue;
  if (a.NB <= 0) return hipSuccess;
  a.M = a.NB * a.H * a.W;
  a.tiles_x = (a.W + IMG_TILE - 1) / IMG_TILE;
  a.tiles_y = (a.H + IMG_TILE - 1) / IMG_TILE;
  a.n_tiles = 1;
  a.m_tiles = a.NB * a.tiles_x * a.tiles_y;
  hipLaunchKernelGGL((conv_igemm_kernel<4, 1, 2, 2, EPI_IMG_STENCIL, PREC>), dim3(a.m_tiles), dim3(256), 0, st, a);
  return hipGetLastError();
}

template <int EPI, int PREC, int TERMS = 7>
inline hipError_t conv_launch_epi(ConvArgs a, hipStream_t st) {
  constexpr int need = PREC != PREC_FP32 ? 7 : 3;                                     // 16 B (fp32) / 32 B (split8) epilogue
  if (PREC == PREC_F16X2 && (EPI == EPI_MUL || EPI == EPI_MUL_UP2) && (!a.tok_fac || !a.tok_max_out || a.out_plain)) return hipErrorInvalidValue;
  if (PREC == PREC_F16X2 && (EPI == EPI_BIAS || EPI == EPI_BIAS_RELU || EPI == EPI_FWD_DUAL) && !a.in_unscale) return hipErrorInvalidValue;
  if ((EPI == EPI_MUL || EPI == EPI_MUL_UP2) && (a.N & need)) return hipErrorInvalidValue;
  if ((EPI == EPI_BIAS || EPI == EPI_BIAS_RELU) && (a.N & 3)) return hipErrorInvalidValue;
  if (PREC != PREC_FP32 && (a.Cin & 7)) return hipErrorInvalidValue;
  if (EPI == EPI_FWD_DUAL && (a.split & 3)) return hipErrorInvalidValue;
  if (EPI == EPI_FWD_DUAL && a.dual_il && ((a.split & 31) || a.N != 2 * a.split)) return hipErrorInvalidValue;
  ConvTile t = conv_pick_tile(a.N);
  if (PREC == PREC_FP32 && t.BN == 128 && (a.N % 64) == 0) {
    // few M rows (the per-image forward at batch 32): 128 x 128 tiles leave CUs idle; halve the tile
    // [MI355X] encode of 32 images 14.06 -> 13.54 ms with the threshold at 2200 blocks (~4 waves of 512 slots)
    constexpr int thr = 2200;
    const long blocks = (((long)a.NB * a.H * a.W + 127) / 128) * ((a.N + 127) / 128);
    if (blocks < thr) t = {128, 64};
  }
  int wide = 0;
  if (PREC != PREC_FP32 && (TERMS == 7 || PREC == PREC_F16X2) &&
      !(PREC == PREC_F16X2 && (EPI == EPI_BIAS || EPI == EPI_BIAS_RELU || EPI == EPI_FWD_DUAL)) &&
      a.N >= 128 && (a.N % 128) == 0) {   // (blocked accumulation does not fit the 8-wave tile)
    wide = conv_wide_tile(a.N, (long)a.NB * a.H * a.W);
    // the tail of a ResNet identity block (1 tap, K = f channels, three full-width streams per output element in the
    // epilogue) is bound by those streams, not by staging: two 128 x 128 workgroups per CU overlap one's epilogue with the
    // other's K loop  [MI355X, ResNet-101 conv4_x: 354 -> 316 us per launch; config 4 -0.9 ms per walk]
    if (EPI == EPI_MUL && a.taps == 1 && a.join) wide = 0;
    if (wide) t = {256, wide};
  }
  a.M = a.NB * a.H * a.W;
  a.m_tiles = (a.M + t.BM - 1) / t.BM;
  a.n_tiles = (a.N + t.BN - 1) / t.BN;
  if (a.M <= 0 || a.N <= 0) return hipSuccess;
  a.epi_generic = sw().epi_fast ? 0 : 1;
  // Small grids (one image, a handful of words: explain_image.py's own call): with 128-row tiles the 14 x 14 / 28 x 28
  // layers are 16-250 workgroups, each walking K = 2304-4608 alone — the launch takes as long as ONE tile's K loop
  // [MI355X, B = 1, T = 10: 115-119 us per block4 / block5 launch, forward and backward].  64 x 64 tiles (4 waves of
  // 32 x 32) put 4x the workgroups on the chip and a k-step costs a quarter of the MFMAs.  Every output element still sees
  // the same chain of MFMAs in the same k order, so the results are bit-identical to the large tiles (batch invariance).
  const bool small_tile = conv_small_tile_on() && t.BM == 128 && t.BN >= 64 && (long)a.m_tiles * a.n_tiles <= conv_small_tile_blocks() &&
                          !a.up2_src && !a.img_part;
  // In between (at most one large tile per CU, but too many for the 64 x 64 tiles to stay resident): 128 x 64 tiles put two
  // workgroups on a CU and halve a k-step — same chains, same bits [MI355X, one image, ten words: block4's walk launches
  // (252 large tiles) 115 -> ~80 us, explain 1.62 -> 1.51 ms].  LRP_CONV_MID=0 disables.
  const bool mid_on = sw().conv_mid != 0;
  if (mid_on && !small_tile && conv_small_tile_on() && t.BM == 128 && t.BN == 128 && (a.N % 64) == 0 && !a.up2_src && !a.img_part &&
      (long)a.m_tiles * a.n_tiles <= 2 * conv_small_tile_blocks()) {
    t = {128, 64};
    a.n_tiles = (a.N + 63) / 64;
  }

  if constexpr (PREC != PREC_FP32 && (EPI == EPI_MUL || EPI == EPI_MUL_UP2 || EPI == EPI_BIAS || EPI == EPI_BIAS_RELU || EPI == EPI_FWD_DUAL)) {
    const int mode = conv_halo_mode();
    if (a.pool_gc) {
      // pool fused into the dual forward's epilogue: 128 x 128 resident-image tiles of even height and width or nothing
      if (!(EPI == EPI_FWD_DUAL && PREC == PREC_F16X2) || !a.dual_il || !a.pairs_out || !a.pool_pos || a.taps != 9 || small_tile ||
          t.BM != 128 || t.BN != 128 || !conv_takes_pool_fused(a.split, a.NB, a.H, a.W))
        return hipErrorInvalidValue;
      (void)conv_halo_geom_even(128, a.H, a.W, a.tw, a.th, a.hrows);
      a.nyh = a.NB * a.H;
      a.cols_t = (a.W + a.tw - 1) / a.tw;
      a.m_tiles = ((a.nyh + a.th - 1) / a.th) * a.cols_t;
      hipLaunchKernelGGL((conv_igemm_kernel<2, 2, 2, 2, EPI, PREC, true, false, TERMS>), dim3(a.m_tiles * a.n_tiles), dim3(256), 0, st, a);
      return hipGetLastError();
    }
    // N = 64 tiles (TM x TN = 2 x 1 per wave) lose with the resident image: 2 instead of 3 blocks per CU and the
    // per-tap address work is spread over half as many MFMAs  [MI355X: block1_conv2 bwd 4.5 ms vs 5.2 ms]
    if constexpr (EPI == EPI_MUL || EPI == EPI_MUL_UP2) {
      // N <= 64: resident image + weights in registers, no barrier per tap (LRP_CONV_BREG=0 disables)
      if (a.taps == 9 && mode > 0 && sw().conv_breg && t.BN == 64 && a.n_tiles == 1 && a.wpk_frag) {
        // (256-row tiles, 8 waves, one block per CU, halve the weight traffic per pixel but lose more to the single
        // block per CU  [MI355X: block1_conv2 bwd 4.19 ms vs 3.83 ms with these 128-row tiles, 3 blocks per CU])
        const float u = conv_halo_geom(t.BM, a.H, a.W, a.tw, a.th, a.hrows);
        if (u >= 0.9f || (mode == 2 && u > 0.f)) {
          a.nyh = a.NB * a.H;
          a.cols_t = (a.W + a.tw - 1) / a.tw;
          a.m_tiles = ((a.nyh + a.th - 1) / a.th) * a.cols_t;
          if (a.img_part) {                              // image layer folded in: tiles laid out per token (none straddles two tokens)
            if (PREC != PREC_BF16X3 || EPI != EPI_MUL || a.N != 64 || !a.img_w) return hipErrorInvalidValue;
            a.tpt = (a.H + a.th - 1) / a.th;
            a.hrows = a.th + 2;                          // (no separator row inside a tile's own rows)
            a.m_tiles = a.NB * a.tpt * a.cols_t;
          }
          a.tile_map = conv_tile_order(a, st);
          if (a.up2_src && (PREC != PREC_BF16X3 || a.CinP > 64 || (!a.up2_gate && !a.up2_pairs) || (a.H & 1) || (a.W & 1))) return hipErrorInvalidValue;
          if (a.up2_pairs && (!a.up2_gpos || !a.img_part)) return hipErrorInvalidValue;   // (pairs need the window loader: per-token tiles = the folded launch)
          hipLaunchKernelGGL((conv_igemm_kernel<2, 2, 2, 1, EPI, PREC, true, true, TERMS>), dim3(a.m_tiles), dim3(256), 0, st, a);
          return hipGetLastError();
        }
      }
    }
    if (a.img_part) return hipErrorInvalidValue;     // the folded image layer exists for the weights-in-registers kernel only
    if (a.up2_src) {
      // compact pool interface into a pipelined halo kernel (ConvArgs::up2_pairs): the launch must reach one (conv_takes_pw)
      if (!((EPI == EPI_MUL || EPI == EPI_MUL_UP2) && PREC == PREC_BF16X3 && TERMS == 7) || !a.up2_pairs || !a.up2_gpos || a.taps != 9 ||
          !conv_takes_pw(a.N, a.NB, a.H, a.W))
        return hipErrorInvalidValue;
    }
    if (small_tile) {
      a.m_tiles = (a.M + 63) / 64;
      a.n_tiles = (a.N + 63) / 64;
      hipLaunchKernelGGL((conv_igemm_kernel<2, 2, 1, 1, EPI, PREC, false, false, TERMS, CONV_SMALL_NS>), dim3(a.m_tiles * a.n_tiles), dim3(256), 0, st, a);
      return hipGetLastError();
    }
    if (a.taps == 9 && mode > 0 && (t.BN >= 128 || (mode == 2 && t.BN >= 64)) && wide != 128) {
      const float u = conv_halo_geom(t.BM, a.H, a.W, a.tw, a.th, a.hrows);
      if (u >= 0.9f || (mode == 2 && u > 0.f)) {
        a.nyh = a.NB * a.H;
        a.cols_t = (a.W + a.tw - 1) / a.tw;
        a.m_tiles = ((a.nyh + a.th - 1) / a.th) * a.cols_t;
        if constexpr (EPI == EPI_MUL || EPI == EPI_MUL_UP2) a.tile_map = conv_tile_order(a, st);
        const dim3 hgrid(a.m_tiles * a.n_tiles);
        if (wide == 256)
          hipLaunchKernelGGL((conv_igemm_kernel<2, 4, 4, 2, EPI, PREC, true, false, TERMS>), hgrid, dim3(512), 0, st, a);
        else if (t.BN == 128)
          hipLaunchKernelGGL((conv_igemm_kernel<2, 2, 2, 2, EPI, PREC, true, false, TERMS>), hgrid, dim3(256), 0, st, a);
        else
          hipLaunchKernelGGL((conv_igemm_kernel<2, 2, 2, 1, EPI, PREC, true, false, TERMS>), hgrid, dim3(256), 0, st, a);
        return hipGetLastError();
      }
    }
  }
  // Only the resident-image kernels above read the compact pool interface.  A launch that carries it and got here (tile
  // override, halo geometry below 0.9 for the tile actually chosen, fp32 operands) would run a kernel that ignores up2_src and
  // reads a.in as a dense tensor: refuse instead of producing wrong heat-maps silently.
  if (a.up2_src || a.pool_gc) return hipErrorInvalidValue;
  if (small_tile) {
    a.m_tiles = (a.M + 63) / 64;
    a.n_tiles = (a.N + 63) / 64;
    hipLaunchKernelGGL((conv_igemm_kernel<2, 2, 1, 1, EPI, PREC, false, false, TERMS, CONV_SMALL_NS>), dim3(a.m_tiles * a.n_tiles), dim3(256), 0, st, a);
    return hipGetLastError();
  }
  const dim3 grid(a.m_tiles * a.n_tiles);
  if constexpr (PREC != PREC_FP32) {
    if (wide == 256) {
      hipLaunchKernelGGL((conv_igemm_kernel<2, 4, 4, 2, EPI, PREC, false, false, TERMS>), grid, dim3(512), 0, st, a);
      return hipGetLastError();
    }
    if (wide == 128) {
      hipLaunchKernelGGL((conv_igemm_kernel<4, 2, 2, 2, EPI, PREC, false, false, TERMS>), grid, dim3(512), 0, st, a);
      return hipGetLastError();
    }
  }
  const dim3 block(256);
  if (t.BN == 128)
    hipLaunchKernelGGL((conv_igemm_kernel<2, 2, 2, 2, EPI, PREC, false, false, TERMS>), grid, block, 0, st, a);
  else if (t.BN == 64)
    hipLaunchKernelGGL((conv_igemm_kernel<2, 2, 2, 1, EPI, PREC, false, false, TERMS>), grid, block, 0, st, a);
  else
    hipLaunchKernelGGL((conv_igemm_kernel<4, 1, 1, 1, EPI, PREC, false, false, TERMS>), grid, block, 0, st, a);
  return hipGetLastError();
}

// prec = PREC_BF16X3 exists for the reverse-walk epilogues (MUL, MUL_UP2, STORE) and the forward Z+ conv (BIAS)
inline hipError_t conv_launch(int epi, const ConvArgs& a, hipStream_t st, int prec = PREC_FP32, int terms = 7) {
  if (prec == PREC_BF16X3 && terms != 7) return hipErrorInvalidValue;   // (the two-pass three-way split forward product of rounds 1-2 is gone)
  if (prec == PREC_F16X2) {                            // reverse walk of the VGG encoder only
    // terms 7: S(hi + lo) x w(hi + lo) without lo*lo — three MFMAs; terms 5: the weights' lo half dropped — two.  Which
    // layers take which is the caller's rule (Encoder::explain: two-term up to the last pool, measured there).
    if (terms == 23) {                                 // interleaved dual forward, Z+ column tiles two-term
      if (epi == EPI_FWD_DUAL && a.dual_il) return conv_launch_epi<EPI_FWD_DUAL, PREC_F16X2, 23>(a, st);
      return hipErrorInvalidValue;
    }
    if (terms == 5) {
      if (epi == EPI_MUL) return conv_launch_epi<EPI_MUL, PREC_F16X2, 5>(a, st);
      if (epi == EPI_MUL_UP2) return conv_launch_epi<EPI_MUL_UP2, PREC_F16X2, 5>(a, st);
      return hipErrorInvalidValue;
    }
    switch (epi) {
      case EPI_MUL: return conv_launch_epi<EPI_MUL, PREC_F16X2>(a, st);
      case EPI_MUL_UP2: return conv_launch_epi<EPI_MUL_UP2, PREC_F16X2>(a, st);
      case EPI_IMG_STENCIL: return conv_launch_img<PREC_F16X2>(a, st);
      case EPI_BIAS_RELU: return conv_launch_epi<EPI_BIAS_RELU, PREC_F16X2>(a, st);   // forward activation conv, fp16 pairs x fp16 pairs
      case EPI_BIAS: return conv_launch_epi<EPI_BIAS, PREC_F16X2>(a, st);
      case EPI_FWD_DUAL: return conv_launch_epi<EPI_FWD_DUAL, PREC_F16X2>(a, st);       // a_l and Z+_l from one pass over x_l
    }
    return hipErrorInvalidValue;
  }
  if (prec == PREC_BF16X3) {
    switch (epi) {
      case EPI_MUL: return conv_launch_epi<EPI_MUL, PREC_BF16X3>(a, st);
      case EPI_MUL_UP2: return conv_launch_epi<EPI_MUL_UP2, PREC_BF16X3>(a, st);
      case EPI_STORE: return conv_launch_epi<EPI_STORE, PREC_BF16X3>(a, st);
      case EPI_IMG_STENCIL: return conv_launch_img<PREC_BF16X3>(a, st);
      case EPI_BIAS: return conv_launch_epi<EPI_BIAS, PREC_BF16X3>(a, st);     // forward Z+ conv (fp32 out)
      case EPI_BIAS_RELU: return conv_launch_epi<EPI_BIAS_RELU, PREC_BF16X3>(a, st);   // forward a conv of the late layers
    }
    return hipErrorInvalidValue;
  }
  switch (epi) {
    case EPI_BIAS_RELU: return conv_launch_epi<EPI_BIAS_RELU, PREC_FP32>(a, st);
    case EPI_BIAS: return conv_launch_epi<EPI_BIAS, PREC_FP32>(a, st);
    case EPI_MUL: return conv_launch_epi<EPI_MUL, PREC_FP32>(a, st);
    case EPI_MUL_UP2: return conv_launch_epi<EPI_MUL_UP2, PREC_FP32>(a, st);
    case EPI_FWD_DUAL: return conv_launch_epi<EPI_FWD_DUAL, PREC_FP32>(a, st);
    case EPI_STORE: return conv_launch_epi<EPI_STORE, PREC_FP32>(a, st);
    case EPI_IMG_STENCIL: return conv_launch_img<PREC_FP32>(a, st);
  }
  return hipErrorInvalidValue;
}

// ---- host-side weight packing (one-off at lrp_set_weight time)
// forward:  wpk[n = co][k = tap*CinP + ci] = w[kh][kw][ci][co]
inline void pack_conv_fwd(const float* w_hwio, int taps, int Cin, int Cout, int col0, int Npad, float* wpk) {
  const int CinP = conv_cinp(Cin), K = taps * CinP;
  (void)Npad;
  for (int t = 0; t < taps; ++t)
    for (int ci = 0; ci < Cin; ++ci)
      for (int co = 0; co < Cout; ++co)
        wpk[(size_t)(col0 + co) * K + t * CinP + ci] = w_hwio[((size_t)t * Cin + ci) * Cout + co];
}
// backward (transposed conv as a forward conv over S with flipped taps):
//   wpk[n = ci][k = tap'*CoutP + co] = w[2-kh'][2-kw'][ci][co]
inline void pack_conv_bwd(const float* w_hwio, int taps, int Cin, int Cout, int col0, float* wpk) {
  const int CoutP = conv_cinp(Cout), K = taps * CoutP;
  for (int t = 0; t < taps; ++t) {
    const int tf = (taps == 9) ? 8 - t : 0;            // (2-kh')*3 + (2-kw') = 8 - t
    for (int ci = 0; ci < Cin; ++ci)
      for (int co = 0; co < Cout; ++co)
        wpk[(size_t)(col0 + ci) * K + t * CoutP + co] = w_hwio[((size_t)tf * Cin + ci) * Cout + co];
  }
}

// fp32 packed matrix [rows][K] (K % 8 == 0) -> split8 in place-compatible layout: per 8 k, 32 B = [hi8 | lo8]
inline unsigned short f32_to_bf16_rne(float f) {
  unsigned u;
  memcpy(&u, &f, 4);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
inline float bf16_to_f32(unsigned short h) {
  unsigned u = (unsigned)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
// BREG operand: split8-packed weights [64 rows][K = taps*CP] -> fragment-major [kc = chunk*9 + tap][step][hi|lo][lane half][64][4 dwords]
inline void pack_frag64(const float* split8_pk, int taps, int CP, float* dst) {
  const int K = taps * CP, cpt = CP / 32;
  for (int cc = 0; cc < cpt; ++cc)
    for (int t = 0; t < taps; ++t) {
      const int kc = cc * taps + t;
      for (int q = 0; q < 4; ++q)
        for (int hh = 0; hh < 2; ++hh)
          for (int n = 0; n < 64; ++n) {
            const int c = 4 * (q >> 1) + 2 * hh + (q & 1);          // 16 B chunk of the 128 B tap-chunk row
            memcpy(dst + ((((size_t)kc * 4 + q) * 2 + hh) * 64 + n) * 4, split8_pk + (size_t)n * K + t * CP + cc * 32 + c * 4, 16);
          }
    }
}

inline void pack_split8(const float* src, size_t n_floats, float* dst_as_float) {
  unsigned short* d = reinterpret_cast<unsigned short*>(dst_as_float);
  for (size_t g = 0; g < n_floats / 8; ++g)
    for (int q = 0; q < 8; ++q) {
      const float x = src[g * 8 + q];
      const unsigned short hi = f32_to_bf16_rne(x);
      d[g * 16 + q] = hi;
      d[g * 16 + 8 + q] = f32_to_bf16_rne(x - bf16_to_f32(hi));
    }
}

#endif  // LRP_CONV_KERNEL_ONLY

}  // namespace lrp
