// conv_sparse.h — the conv-LRP launch BEHIND a 2x2 max-pool on the 2:4-sparse matrix cores (v_smfmac_f32_32x32x32_bf16).
//
// The relevance that enters block4_conv3 / block3_conv3 (/ block2_conv2 / block1_conv2) came through a max-pool (RA:470-480 ->
// IL:138-157): S[n][Y][X][c] = S_c[n][Y/2][X/2][c] where the window's arg-max sits, zero on the three other pixels.  The dense
// kernel (conv_igemm.h) multiplies those zeros: nine taps x C channels per output pixel of which at most four carry a value.
// Grouped by the 2x2 WINDOW a tap falls into, the nine taps of an output pixel of parity class q = (qy, qx) are
//     own window        4 taps   <= 1 non-zero
//     horizontal nbr    2 taps   <= 1 non-zero        (the window at wx + sx, sx = qx ? +1 : -1; only its near column)
//     vertical nbr      2 taps   <= 1 non-zero        (wy + sy; only its near row)
//     diagonal nbr      1 tap    <= 1 non-zero
// so three groups of four k-slots — (own 0..3), (h row 0, h row 1, diagonal, dump), (v col 0, v col 1, dump, dump) — are each
// 2:4 sparse BY CONSTRUCTION: 12 slots per channel at the sparse rate = 6 dense-equivalent k instead of 9 (1.5x fewer matrix
// cycles, x 3 split-bf16 products as before).  The A operand needs no masking: the compressed values are the neighbour windows'
// S_c themselves and the 2-bit indices say where they sit — a value whose position is not adjacent to the output pixel is
// pointed at a DUMP slot whose weight row is zero.  (profiles/smfmac_semantics.hip: on gfx950 the index pairs of a group need
// not be ordered and may even coincide; lane layout of the compressed operand found there.)
//
// What differs per parity class is which taps the slots stand for, i.e. the B matrix: a tile therefore holds output pixels of
// ONE class (rows = windows), four weight arrangements exist (4/3 of the dense bytes each), and the grid walks class by class
// so that the workgroups resident on an XCD share one arrangement in its L2.
//
// Tile 256 windows x 256 output channels, 8 waves of 128 x 64 (as the dense 8-wave tile); per 8-channel set three k-steps
// (own | h + d | v) of 24 smfmacs per wave, two k-steps per barrier.  B: LDS-DMA, four 32 KB stages (two super-steps).
// A: the tile's windows + a one-window ring, 96 B per window and chunk (pairs in lane order + four planes of index words, the
// latter precomputed once per image by conv_sparse_index_kernel), copied through registers a step ahead, double-buffered.
#pragma once
#include "cnn_kernels.h"
#include "conv_igemm.h"

namespace lrp {

typedef __bf16 bf16x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

struct SparseArgs {
  const float* sc;            // S_c at pooled resolution as split8 pairs, CHUNK-MAJOR: [NB][C / 16][Hp][Wp][set 0: hi8 | lo8, set 1: hi8 | lo8]
                              // (64 B per window and 16-channel chunk: what a tile stages per chunk is contiguous; conv_sparse_pairs_kernel)
  const unsigned* idxp;       // conv_sparse_index_kernel's output (from the arg-max positions 2 dy + dx of the pool): per image, parity
                              // class, chunk and window 2 x 16 B of index words [images][4 classes][C / 16][Hp][Wp][2 sets][2 halves][own|h|d|v]
  const float* wsp;           // conv_sparse_pack_kernel's output: [4 classes][n_tiles][C / 16][6 steps][32 KB]
  const float* gate;          // G of the layer below [images][2 Hp][2 Wp][N] fp32
  float* out;                 // [NB][2 Hp][2 Wp][N] split8 pairs (fp32 when out_plain)
  const int* row2img;         // token -> image (nullptr = identity)
  int NB, Hp, Wp, C, N;
  int cols_t, m_tiles, n_tiles;
  int out_plain;
  int diag;                   // measurement only (profiles/sparse_ab.py): bit 0 no epilogue stores
};

constexpr int SP_TW = 14, SP_TH = 18, SP_PITCH = SP_TW + 2;      // tile = 18 stack window rows x 14 window columns = 252 rows
constexpr int SP_NENT = (SP_TH + 2) * SP_PITCH;                  // resident windows (tile + ring); entry SP_NENT = all-zero
constexpr int SP_PENT = SP_NENT + 1;                             // entries of a sub-plane (the last one is the all-zero entry)
constexpr int SP_SUB = SP_PENT * 16;                             // bytes of a sub-plane: 16 B per resident window = [lane half 0: 8 B | half 1: 8 B]
constexpr int SP_ABUF = 3 * SP_SUB;                              // one 8-channel set: [pairs hi | pairs lo | index words]
constexpr int SP_SUBPIECES = (SP_PENT + 63) / 64;                // 1 KiB DMA pieces per sub-plane (the last one overlaps the one before)
constexpr int SP_APIECES = 3 * SP_SUBPIECES;
constexpr int SP_ASLOTS = (SP_APIECES + 7) / 8;                  // pieces per wave and set
constexpr int SP_BSTAGE = 32768, SP_NSTAGE = 4;
constexpr int SP_LDS = SP_NSTAGE * SP_BSTAGE + 2 * SP_ABUF + 128;      // (+ the window row -> image table)
static_assert(SP_LDS <= 160 * 1024, "LDS");
static_assert(SP_PENT >= 64, "a DMA piece is 64 windows");

// B operand of class q: slot -> tap of the backward conv (the matrix conv_igemm's launch multiplies with: row = output channel
// ci, k = tap * CPo + co, tap = 3 (dy + 1) + (dx + 1) reads S[y + dy][x + dx]); -1 = dump slot (zero row).
__host__ __device__ inline int sparse_slot_tap(int q, int group, int slot) {
  const int qy = q >> 1, qx = q & 1, sy = qy ? 1 : -1, sx = qx ? 1 : -1;
  int dy, dx;
  if (group == 0) { dy = (slot >> 1) - qy; dx = (slot & 1) - qx; }             // own window: slot = position 2 py + px
  else if (group == 1) {
    if (slot < 2) { dy = slot - qy; dx = sx; }                                  // horizontal neighbour, its near column: slot = py
    else if (slot == 2) { dy = sy; dx = sx; }                                   // diagonal neighbour, its near corner
    else return -1;
  } else {
    if (slot < 2) { dy = sy; dx = slot - qx; }                                  // vertical neighbour, its near row: slot = px
    else return -1;
  }
  return 3 * (dy + 1) + (dx + 1);
}

// Index words of the compressed A operand, once per IMAGE (the positions belong to the image, not to the token; round 4 first
// computed them while staging every tile: ~80 VALU instructions per item).  pos [images][Hp][Wp][C] bytes = 2 py + px.
// Per (window, 8 channels, class q) four planes x two lane halves of 16-bit words, one 2-bit-pair nibble per channel:
//   own   (pos, pos ^ 1)                         the second value of the pair is zero
//   h     low index  = py, or dump slot 3        valid iff the non-zero sits in the column next to a class-q pixel (px != qx)
//   d     high index = slot 2, or dump slot 3    valid iff it sits in the near corner
//   v     (px or dump slot 2, 3)                 valid iff it sits in the near row (py != qy); second value zero
// Lane half h of a fragment holds channels 4h .. 4h + 3 of the set (the B operand's k order is permuted to match: the compressed
// operand's own k order would pair {2h, 2h+1, 4+2h, 5+2h}, profiles/smfmac_semantics.hip).
__global__ __launch_bounds__(256) void conv_sparse_index_kernel(const unsigned char* __restrict__ pos, unsigned* __restrict__ idxp, size_t n_sets,
                                                                int HW, int C) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_sets * 4; i += (size_t)gridDim.x * 256) {
    const int q = (int)(i & 3);
    const size_t set = i >> 2;                             // (image, window, set of 8 channels) in the positions' own order
    const u32x2 pp = *reinterpret_cast<const u32x2*>(pos + set * 8);
    const unsigned qxm = (q & 1) ? 0x01010101u : 0u, qym = (q >> 1) ? 0x01010101u : 0u;
    unsigned word[4][2];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const unsigned P = pp[d];
      const unsigned px = P & 0x01010101u, py = (P >> 1) & 0x01010101u;
      const unsigned mh = px ^ qxm, mv = py ^ qym;
      const unsigned nmh = mh ^ 0x01010101u, nmv = mv ^ 0x01010101u;
      word[0][d] = P | ((P ^ 0x01010101u) << 2);
      word[1][d] = py | nmh | (nmh << 1);
      word[2][d] = (0x02020202u | ((mh & mv) ^ 0x01010101u)) << 2;
      word[3][d] = (px & mv) | (nmv << 1) | 0x0C0C0C0Cu;
    }
    unsigned h0[4], h1[4];                                   // per plane: the four nibbles of channels 0..3 / 4..7
#pragma unroll
    for (int pl = 0; pl < 4; ++pl) {
      const unsigned t0 = word[pl][0] | (word[pl][0] >> 4), t1 = word[pl][1] | (word[pl][1] >> 4);
      h0[pl] = (t0 & 0xFFu) | (((t0 >> 16) & 0xFFu) << 8);
      h1[pl] = (t1 & 0xFFu) | (((t1 >> 16) & 0xFFu) << 8);
    }
    const u32x4 iw = {h0[0] | (h0[1] << 16), h0[2] | (h0[3] << 16), h1[0] | (h1[1] << 16), h1[2] | (h1[3] << 16)};
    const int spw = C >> 3;                                  // sets per window
    const size_t win = set / spw;
    const int sw = (int)(set - win * spw), chunk = sw >> 1, s2 = sw & 1;
    const size_t img = win / HW, pix = win - img * HW;
    reinterpret_cast<u32x4*>(idxp)[((((img * 4 + q) * (C >> 4) + chunk) * HW + pix) << 1) + s2] = iw;
  }
}

// S_c fp32 [NB][Hp * Wp][C] -> split8 pairs in the chunk-major layout the tiles stage from: [NB][C / 16][Hp * Wp][2 sets x (hi8 | lo8)]
__global__ __launch_bounds__(256) void conv_sparse_pairs_kernel(const float* __restrict__ sc, float* __restrict__ scp, size_t n_sets, int HW, int C) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_sets; i += (size_t)gridDim.x * 256) {
    float v[8];
    *reinterpret_cast<f32x4*>(v) = *reinterpret_cast<const f32x4*>(sc + i * 8);
    *reinterpret_cast<f32x4*>(v + 4) = *reinterpret_cast<const f32x4*>(sc + i * 8 + 4);
    const int spw = C >> 3;
    const size_t win = i / spw;
    const int sw = (int)(i - win * spw), chunk = sw >> 1, s2 = sw & 1;
    const size_t n = win / HW, pix = win - n * HW;
    split8_store(v, scp + ((((n * (C >> 4) + chunk) * HW + pix) << 1) + s2) * 8);
  }
}

// ... and the same re-layout of pairs the walk's producer has already written window-major ([NB][Hp * Wp][C] split8 groups)
__global__ __launch_bounds__(256) void conv_sparse_relayout_kernel(const float* __restrict__ pairs, float* __restrict__ scp, size_t n_sets, int HW, int C) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_sets; i += (size_t)gridDim.x * 256) {
    const u32x4 hi = *reinterpret_cast<const u32x4*>(pairs + i * 8), lo = *reinterpret_cast<const u32x4*>(pairs + i * 8 + 4);
    const int spw = C >> 3;
    const size_t win = i / spw;
    const int sw_ = (int)(i - win * spw), chunk = sw_ >> 1, s2 = sw_ & 1;
    const size_t n = win / HW, pix = win - n * HW;
    float* dst = scp + ((((n * (C >> 4) + chunk) * HW + pix) << 1) + s2) * 8;
    *reinterpret_cast<u32x4*>(dst) = hi;
    *reinterpret_cast<u32x4*>(dst + 4) = lo;
  }
}

// wsp[q][nt][chunk][step][hi|lo][lane half][part][256 cols][8 bf16] from the fp32 backward matrix wb [Npad][9 * CPo].
// Lane (col, half) of a B fragment holds k = 16 half + 0..15 = channels 4 half .. 4 half + 3 of the step's eight, 4 slots each.
__global__ __launch_bounds__(256) void conv_sparse_pack_kernel(const float* __restrict__ wb, float* __restrict__ wsp, int N, int C, int CPo,
                                                               int n_tiles) {
  const size_t total = (size_t)4 * n_tiles * (C / 16) * 6 * 2 * 2 * 256;           // (hi and lo written by one thread)
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int col = (int)(i & 255);
    size_t r = i >> 8;
    const int part = (int)(r & 1); r >>= 1;
    const int half = (int)(r & 1); r >>= 1;
    const int step = (int)(r % 6); r /= 6;
    const int chunk = (int)(r % (C / 16)); r /= (C / 16);
    const int nt = (int)(r % n_tiles);
    const int q = (int)(r / n_tiles);
    const int ci = nt * 256 + col;
    bf16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      // B's k = 4 cb + slot; the A lane (half h, group g) meets cb = 4 (g / 2) + 2 h + (g % 2) and holds real channel 4 h + g
      const int k32 = 16 * half + 8 * part + e, cb = k32 >> 2, slot = k32 & 3;
      const int ch8 = 4 * ((cb >> 1) & 1) + 2 * (cb >> 2) + (cb & 1);
      const int co = chunk * 16 + (step & 1) * 8 + ch8;
      const int tap = sparse_slot_tap(q, step >> 1, slot);
      const float v = (tap >= 0 && ci < N && co < C) ? wb[(size_t)ci * 9 * CPo + (size_t)tap * CPo + co] : 0.f;
      hi[e] = (__bf16)v;
      lo[e] = (__bf16)(v - (float)hi[e]);
    }
    // element offset (in 16 B units) inside the step's 32 KB: [hl][half][part][col]
    const size_t sbase = ((((size_t)q * n_tiles + nt) * (C / 16) + chunk) * 6 + step) * (SP_BSTAGE / 16);
    u32x4* dst = reinterpret_cast<u32x4*>(wsp);
    dst[sbase + ((0 * 2 + half) * 2 + part) * 256 + col] = __builtin_bit_cast(u32x4, hi);
    dst[sbase + ((1 * 2 + half) * 2 + part) * 256 + col] = __builtin_bit_cast(u32x4, lo);
  }
}

__global__ __launch_bounds__(512, 2) void conv_sparse_kernel(SparseArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(16))) unsigned char lds[SP_LDS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 2, wn = wave & 3;
  const int l31 = lane & 31, half = lane >> 5;
  const int per_class = a.m_tiles * a.n_tiles;
  const int q = blockIdx.x / per_class, rem = blockIdx.x - q * per_class;
  const int nt = rem / a.m_tiles, mt = rem - nt * a.m_tiles;
  const int qy = q >> 1, qx = q & 1, sy = qy ? 1 : -1, sx = qx ? 1 : -1;
  const int tyt = mt / a.cols_t, txt = mt - tyt * a.cols_t;
  const int Y0 = tyt * SP_TH, x0 = txt * SP_TW;            // first stack window row / window column of the tile
  const int Hp = a.Hp, Wp = a.Wp, C = a.C, N = a.N, H = 2 * Hp, W = 2 * Wp;
  const int nys = a.NB * Hp;                                // window rows of the stack
  const int n0 = nt * 256;
  const float inv_Hp = 1.0f / (float)Hp;
  auto divmod = [](int x, int d, float inv, int& qq, int& r) {
    qq = (int)(((float)x + 0.5f) * inv);
    r = x - qq * d;
    if (r < 0) { --qq; r += d; } else if (r >= d) { ++qq; r -= d; }
  };
  unsigned char* Bs = lds;
  unsigned char* As = lds + SP_NSTAGE * SP_BSTAGE;

  // ---- A in LDS: six sub-planes per buffer — pairs hi, pairs lo, index words, each for set 0 and set 1 — of 16 B per resident
  // window: [lane half 0: 8 B | lane half 1: 8 B].  The 16 lanes LDS serves per cycle of a ds_read_b64 read 16 (nearly) consecutive
  // windows at a 16 B pitch: all 64 banks once, no swizzle needed; and a sub-plane is a plain 16 B-per-lane LDS-DMA image of what
  // lies in global memory.  Per A fragment: the sub-plane offsets of the four windows its row reads — own, horizontal, vertical,
  // diagonal neighbour; a neighbour outside the image, or in another token, is the all-zero entry.  Fixed for the whole K loop.
  auto chunk0 = [&](int e) { return e * 16 + half * 8; };
  int e_own[4], e_h[4], e_v[4], e_d[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (wm * 4 + i) * 32 + l31;
    const int ty = r / SP_TW, tx = r - ty * SP_TW;
    const int Ys = Y0 + ty, wx = x0 + tx;
    int n_, wy;
    divmod(Ys, Hp, inv_Hp, n_, wy);
    const bool ok = r < SP_TH * SP_TW && Ys < nys && wx < Wp;
    const bool okh = ok && wx + sx >= 0 && wx + sx < Wp, okv = ok && wy + sy >= 0 && wy + sy < Hp;
    const int e = (ty + 1) * SP_PITCH + tx + 1;
    e_own[i] = chunk0(ok ? e : SP_NENT);
    e_h[i] = chunk0(okh ? e + sx : SP_NENT);
    e_v[i] = chunk0(okv ? e + sy * SP_PITCH : SP_NENT);
    e_d[i] = chunk0(okh && okv ? e + sy * SP_PITCH + sx : SP_NENT);
  }

  // ---- B: LDS-DMA, 4 x 1 KiB per wave and step
  typedef __attribute__((address_space(3))) void* lptr_t;
  constexpr int RSRC_FLAGS = 0x00020000;
  const int nchunks = C >> 4, nsteps = nchunks * 6;
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(reinterpret_cast<const unsigned char*>(a.wsp) + ((size_t)q * a.n_tiles + nt) * (size_t)nsteps * SP_BSTAGE), 0, nsteps * SP_BSTAGE, RSRC_FLAGS);
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  auto fire_b = [&](int step, int stage) {
#pragma unroll
    for (int p = 0; p < 4; ++p)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lptr_t)(Bs + stage * SP_BSTAGE + (wave_s * 4 + p) * 1024), 16, lane * 16,
                                               step * SP_BSTAGE + (wave_s * 4 + p) * 1024, 0, 0);
  };

  // ---- A staging: by LDS-DMA as well, 16 B per lane: a sub-plane entry IS 16 contiguous bytes of global memory — the hi (or lo)
  // half of a window's split8 group of the set, or the index words conv_sparse_index_kernel left for (window, set, class) — so no
  // register, no VALU on the data, and no register load shares vmcnt with the B stages.  (Earlier forms, all measured [MI355X,
  // block4_conv3, 2.5 ms launch]: through registers, +0.4 ms — hipcc waits vmcnt(0) for a register load that has LDS-DMA behind it
  // on the counter, and a load that misses L2 stalls the step it was issued in, also when only half of the waves stage; one DWORD
  // per lane into a swizzled 32 B-per-window image, +0.9 ms — 123 DMA instructions per chunk.)  A piece = 64 windows of one
  // sub-plane; wave w moves pieces w, w + 8, ...; windows outside the image and the all-zero entry are out-of-range offsets = zeros.
  int* imgtab = reinterpret_cast<int*>(As + 2 * SP_ABUF);       // resident window row hy -> image of its token
  if (tid < SP_TH + 2) {
    const int Ys = Y0 - 1 + tid;
    int n_, wy;
    divmod(Ys >= 0 && Ys < nys ? Ys : 0, Hp, inv_Hp, n_, wy);
    imgtab[tid] = a.row2img ? a.row2img[n_] : n_;
  }
  const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc((void*)a.sc, 0, 0x7FFFFFFF, RSRC_FLAGS);
  const __amdgpu_buffer_rsrc_t rsI = __builtin_amdgcn_make_buffer_rsrc((void*)a.idxp, 0, 0x7FFFFFFF, RSRC_FLAGS);
  constexpr int OOB = (int)0x80000000;
  auto fire_a = [&](int slot, int set8, int abuf) {           // set8 = 8-channel set of the layer (two per 16-channel chunk of the sources)
    // (opaque copies: the address math of a slot does not depend on the set, and hipcc otherwise keeps every slot's worth of it
    //  alive across the K loop; recomputed per call it is ~25 VALU instructions)
    int wv = wave_s, ln = lane;
    asm volatile("" : "+s"(wv), "+v"(ln));
    int pj = wv + 8 * slot;
    if (pj >= SP_APIECES) pj = SP_APIECES - 1;               // (spare slots repeat the last piece: idempotent)
    const int pl = pj / SP_SUBPIECES, jp = pj - pl * SP_SUBPIECES;         // sub-plane (hi, lo, index words), piece of it (wave-uniform)
    const int first = jp * 64 < SP_PENT - 64 ? jp * 64 : SP_PENT - 64;     // (the last piece overlaps its predecessor)
    const int e = first + ln;
    const int hy = e >> 4, hx = e & 15;
    const int Ys = Y0 - 1 + hy, wx = x0 - 1 + hx;
    const bool ok = e < SP_NENT && Ys >= 0 && Ys < nys && wx >= 0 && wx < Wp;
    int n_, wy;
    divmod(ok ? Ys : 0, Hp, inv_Hp, n_, wy);
    const int chunk = set8 >> 1, s = set8 & 1;
    int vo;
    if (pl < 2) {
      vo = ((((n_ * (C >> 4) + chunk) * Hp + wy) * Wp + wx) << 6) + s * 32 + pl * 16;       // the set's hi8 (pl 0) or lo8 (pl 1)
    } else {
      const int img = imgtab[ok ? hy : 0];
      vo = (((((img * 4 + q) * (C >> 4) + chunk) * Hp + wy) * Wp + wx) << 5) + s * 16;
    }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(pl < 2 ? rsS : rsI, (lptr_t)(As + abuf * SP_ABUF + pl * SP_SUB + first * 16), 16, ok ? vo : OOB,
                                             0, 0, 0);
  };
  auto fire_set = [&](int set8, int abuf) {
#pragma unroll
    for (int sl = 0; sl < SP_ASLOTS; ++sl) fire_a(sl, set8, abuf);
  };

  // ---- K loop.  A set = 8 channels = three k-steps (own | h + d | v); the sources are 16-channel chunks = two sets = six steps,
  // walked as three SUPER-STEPS of two k-steps with ONE barrier each (48 smfmacs per wave between barriers, like the dense tile):
  //     super-step 0: (set 0 own, set 0 h+d)    1: (set 0 v, set 1 own)    2: (set 1 h+d, set 1 v)
  // B: four 32 KB stages = two super-steps; at the top of a super-step the two stages of the NEXT one are requested (into the
  // stages the previous one just released).  A: one buffer per set parity; set 0 of the next chunk is requested at the top of
  // super-step 2 (its buffer was last read in super-step 1), set 1 at the top of the next chunk's super-step 0.  Everything
  // requested at the top of a super-step has landed by its end: one s_waitcnt vmcnt(0) per barrier, nothing to count.
  const int nsteps2 = nchunks * 6;
  // global step index t -> where its 32 KB of B lie: chunk16 * 6 + 2 * group + set  (the pack kernel's order)
  auto bsrc = [&](int chunk, int t6) { return chunk * 6 + 2 * (t6 % 3) + t6 / 3; };
  __syncthreads();                                         // (imgtab)
  fire_set(0, 0);
  if (nchunks * 2 > 1) fire_set(1, 1);
  fire_b(bsrc(0, 0), 0);
  fire_b(bsrc(0, 1), 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int bcol = (wn * 64 + l31) * 16 + half * 8192;
  auto expand = [](u32x2 x) -> bf16x8 {                     // [a|b], [c|d] -> (a, 0), (b, 0), (c, 0), (d, 0)
    const u32x4 r = {x[0] & 0xFFFFu, x[0] >> 16, x[1] & 0xFFFFu, x[1] >> 16};
    return __builtin_bit_cast(bf16x8, r);
  };
  auto pairup = [](u32x2 h, u32x2 d) -> bf16x8 {           // (h_a, d_a), (h_b, d_b), (h_c, d_c), (h_d, d_d)
    const u32x4 r = {__builtin_amdgcn_perm(d[0], h[0], 0x05040100u), __builtin_amdgcn_perm(d[0], h[0], 0x07060302u),
                     __builtin_amdgcn_perm(d[1], h[1], 0x05040100u), __builtin_amdgcn_perm(d[1], h[1], 0x07060302u)};
    return __builtin_bit_cast(bf16x8, r);
  };
  auto ld8 = [](const unsigned char* p) { return *reinterpret_cast<const u32x2*>(p); };
  auto ld2 = [](const unsigned char* p) { return (unsigned)*reinterpret_cast<const unsigned short*>(p); };
  struct BF { bf16x16 h[2], l[2]; };
  auto load_b = [&](BF& b, const unsigned char* Bb) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const unsigned char* bp = Bb + j * 512;                               // [hi|lo][half][part][col] x 16 B
      const bf16x8 h0 = *reinterpret_cast<const bf16x8*>(bp), h1 = *reinterpret_cast<const bf16x8*>(bp + 4096);
      const bf16x8 l0 = *reinterpret_cast<const bf16x8*>(bp + 16384), l1 = *reinterpret_cast<const bf16x8*>(bp + 16384 + 4096);
      b.h[j] = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
      b.l[j] = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    }
  };
  // one k-step: group grp (0 own, 1 h + d, 2 v) of the set in buffer Ab against the B fragments b
  auto kstep = [&](const unsigned char* Ab, int grp, const BF& b) {
    const unsigned char* Ah = Ab;
    const unsigned char* Al = Ab + SP_SUB;
    const unsigned char* Ai = Ab + 2 * SP_SUB;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      bf16x8 vh, vl;
      int ix;
      if (grp == 1) {
        const int oh = e_h[i], od = e_d[i];
        vh = pairup(ld8(Ah + oh), ld8(Ah + od));
        vl = pairup(ld8(Al + oh), ld8(Al + od));
        ix = (int)(ld2(Ai + oh + 2) | ld2(Ai + od + 4));
      } else {
        const int oo = grp == 0 ? e_own[i] : e_v[i];
        vh = expand(ld8(Ah + oo));
        vl = expand(ld8(Al + oo));
        ix = (int)ld2(Ai + oo + (grp == 0 ? 0 : 6));
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        acc[i][j] = __builtin_amdgcn_smfmac_f32_32x32x32_bf16(vl, b.h[j], acc[i][j], ix, 0, 0);     // small terms first
        acc[i][j] = __builtin_amdgcn_smfmac_f32_32x32x32_bf16(vh, b.l[j], acc[i][j], ix, 0, 0);
        acc[i][j] = __builtin_amdgcn_smfmac_f32_32x32x32_bf16(vh, b.h[j], acc[i][j], ix, 0, 0);
      }
    }
  };

  for (int chunk = 0; chunk < nchunks; ++chunk) {
    const bool next_chunk = chunk + 1 < nchunks;
#pragma unroll
    for (int ss = 0; ss < 3; ++ss) {
      const int t0 = 2 * ss, t1 = 2 * ss + 1;              // the super-step's two k-steps of the chunk's six
      const int st0 = chunk * 6 + t0;                        // global step index: its B stage is st % 4
      // (1) requests: the next super-step's two B stages, and the A set whose buffer has just been released
      if (st0 + 2 < nsteps2) {
        const int nc = ss == 2 ? chunk + 1 : chunk, nt0 = (t0 + 2) % 6;
        fire_b(bsrc(nc, nt0), (st0 + 2) & 3);
        fire_b(bsrc(nc, nt0 + 1), (st0 + 3) & 3);
      }
      if (ss == 2 && next_chunk) fire_set(2 * chunk + 2, 0);
      if (ss == 0 && chunk > 0) fire_set(2 * chunk + 1, 1);
      // (2) the two k-steps; the second one's B fragments are read while the first one's smfmacs run
      BF b0, b1;
      load_b(b0, Bs + (st0 & 3) * SP_BSTAGE + bcol);
      load_b(b1, Bs + ((st0 + 1) & 3) * SP_BSTAGE + bcol);
      kstep(As + (t0 / 3) * SP_ABUF, t0 % 3, b0);
      kstep(As + (t1 / 3) * SP_ABUF, t1 % 3, b1);
      // (3) everything requested at the top has landed; everyone is done with this super-step's stages and (ss 1, 2) A buffer.
      // (A bare s_barrier: __syncthreads() adds hipcc's workgroup fence.)
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
  }

  // ---- epilogue: out = acc x gate at the class-q pixel of every window, 64 rows of the tile at a time through LDS
  float* Cs = reinterpret_cast<float*>(lds);               // 64 x 256 fp32 = 64 KB (the B stages)
#pragma unroll
  for (int sl = 0; sl < 4; ++sl) {
    // the slab's gate rows first: their latency runs under the C tile's trip through LDS
    f32x4 g[4][2];
    bool ok4[4];
    size_t off4[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int item = tid + p * 512, row = item >> 5, cg = item & 31;
      const int R = sl * 64 + row;
      const int ty = R / SP_TW, tx = R - ty * SP_TW;
      const int Ys = Y0 + ty, wx = x0 + tx;
      int n_, wy;
      divmod(Ys < nys ? Ys : 0, Hp, inv_Hp, n_, wy);
      const int col = n0 + cg * 8;
      ok4[p] = R < SP_TH * SP_TW && Ys < nys && wx < Wp && col < N && !((a.diag & 1) && (row | cg | sl));
      const int y = 2 * wy + qy, x = 2 * (wx < Wp ? wx : 0) + qx;
      const int img = a.row2img ? a.row2img[n_] : n_;
      const size_t goff = (((size_t)img * H + y) * W + x) * N + (col < N ? col : 0);
      off4[p] = (((size_t)n_ * H + y) * W + x) * N + (col < N ? col : 0);
      g[p][0] = *reinterpret_cast<const f32x4*>(a.gate + goff);
      g[p][1] = *reinterpret_cast<const f32x4*>(a.gate + goff + 4);
    }
    if (sl) __syncthreads();
    if (wm == (sl >> 1)) {
      float* cw = Cs + (4 * half) * 256 + wn * 64 + l31;
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
          for (int j = 0; j < 2; ++j) cw[(ii * 32 + (r & 3) + 8 * (r >> 2)) * 256 + j * 32] = acc[2 * (sl & 1) + ii][j][r];
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      if (!ok4[p]) continue;
      const int item = tid + p * 512, row = item >> 5, cg = item & 31;
      const f32x4 c0 = *reinterpret_cast<const f32x4*>(Cs + row * 256 + cg * 8), c1 = *reinterpret_cast<const f32x4*>(Cs + row * 256 + cg * 8 + 4);
      float r[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) { r[e] = c0[e] * g[p][0][e]; r[4 + e] = c1[e] * g[p][1][e]; }
      float* dst = a.out + off4[p];
      if (a.out_plain) {
        *reinterpret_cast<f32x4*>(dst) = *reinterpret_cast<const f32x4*>(r);
        *reinterpret_cast<f32x4*>(dst + 4) = *reinterpret_cast<const f32x4*>(r + 4);
      } else {
        split8_store(r, dst);
      }
    }
  }
#endif
}

// floats of the packed weights of a layer (all four classes)
inline size_t conv_sparse_weight_floats(int N, int C) { return (size_t)4 * ((N + 255) / 256) * (C / 16) * 6 * (SP_BSTAGE / 4); }
inline bool conv_sparse_supports(int N, int C, int Hp, int Wp) { return N % 256 == 0 && C % 16 == 0 && Hp >= 1 && Wp >= 1; }

inline hipError_t conv_sparse_pack(const float* wb_dev, float* wsp_dev, int N, int C, hipStream_t st) {
  const int n_tiles = (N + 255) / 256;
  const size_t total = (size_t)4 * n_tiles * (C / 16) * 6 * 2 * 2 * 256;
  hipLaunchKernelGGL(conv_sparse_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, wb_dev, wsp_dev, N, C, conv_cinp(C), n_tiles);
  return hipGetLastError();
}

// index planes of a layer's pool positions: floats of the buffer, and the pre-pass itself (once per encode and pooled layer)
inline size_t conv_sparse_index_words(int images, int Hp, int Wp, int C) { return (size_t)images * Hp * Wp * (C / 8) * 4 * 4; }
inline hipError_t conv_sparse_index(const unsigned char* pos_dev, unsigned* idxp_dev, int images, int Hp, int Wp, int C, hipStream_t st) {
  const size_t n_sets = (size_t)images * Hp * Wp * (C / 8);
  hipLaunchKernelGGL(conv_sparse_index_kernel, dim3(stream_grid(n_sets * 4)), dim3(256), 0, st, pos_dev, idxp_dev, n_sets, Hp * Wp, C);
  return hipGetLastError();
}
// S_c fp32 (NB, Hp, Wp, C) -> the chunk-major pairs a launch reads (NB * Hp * Wp * C floats)
inline hipError_t conv_sparse_pairs(const float* sc_dev, float* scp_dev, int NB, int Hp, int Wp, int C, hipStream_t st) {
  const size_t n_sets = (size_t)NB * Hp * Wp * (C / 8);
  hipLaunchKernelGGL(conv_sparse_pairs_kernel, dim3(stream_grid(n_sets)), dim3(256), 0, st, sc_dev, scp_dev, n_sets, Hp * Wp, C);
  return hipGetLastError();
}

inline hipError_t conv_sparse_launch(SparseArgs a, hipStream_t st) {
  if (!conv_sparse_supports(a.N, a.C, a.Hp, a.Wp) || !a.sc || !a.idxp || !a.wsp || !a.gate || !a.out) return hipErrorInvalidValue;
  if (a.NB <= 0) return hipSuccess;
  a.cols_t = (a.Wp + SP_TW - 1) / SP_TW;
  a.n_tiles = a.N / 256;
  a.m_tiles = ((a.NB * a.Hp + SP_TH - 1) / SP_TH) * a.cols_t;
  hipLaunchKernelGGL(conv_sparse_kernel, dim3(4 * a.m_tiles * a.n_tiles), dim3(512), 0, st, a);
  return hipGetLastError();
}

}  // namespace lrp
