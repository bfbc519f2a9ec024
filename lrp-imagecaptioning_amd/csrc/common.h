// common.h — error plumbing and small RAII helpers shared by the engine sources.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/lrp_hip.h"

namespace lrp {

// Every kernel launch of the library goes through hipLaunchKernelGGL; this wrapper counts them (lrp_launch_count: bench.py
// reports the launches behind one single-image explanation next to its latency).  Memsets / copies are not counted.
// (atomic, relaxed: handles may be driven from several host threads — bench.py's power probe does)
inline std::atomic<unsigned long long> g_launch_count{0};
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernelName, numBlocks, numThreads, memPerBlock, streamId, ...)                  \
  do {                                                                                                    \
    ::lrp::g_launch_count.fetch_add(1, std::memory_order_relaxed);                                       \
    kernelName<<<(numBlocks), (numThreads), (memPerBlock), (streamId)>>>(__VA_ARGS__);                   \
  } while (0)

std::string& last_error_ref();

// A/B switches kept for measurement and as tested fall-backs (every one of them is exercised by
// tests/test_gpu_switches.py against the default path).  Read from the environment ONCE per process — no getenv on the launch
// path — and again by lrp_reload_switches() (include/lrp_hip.h), which is how a test or a measurement script flips one.
struct Switches {
  int conv_halo = 1;        // LRP_CONV_HALO  0 never / 1 when a tile shape fills >= 90 % of the M tile / 2 always (ragged shapes: tests)
  int conv_breg = 1;        // LRP_CONV_BREG=0     N <= 64 backward convs without the weights-in-registers kernel
  int conv_tile = 0;        // LRP_CONV_TILE  0 auto / 1 never the 8-wave tiles / 128 cap them at 256 x 128
  int conv_small = 1;       // LRP_CONV_SMALL=0    small grids keep the 128-row tiles (no 64 x 64 tiles)
  int conv_mid = 1;         // LRP_CONV_MID=0      no 128 x 64 tiles for the grids just above the small ones
  int epi_fast = 1;         // LRP_EPI_FAST=0      MUL / MUL_UP2 epilogues always through the general pass loop
  int up2_pw = 1;           // LRP_UP2_PW=0        pooled boundaries of the pipelined halo kernels through the expanded tensor
  int tile_order = 1;       // LRP_TILE_ORDER=0    reverse-walk launches in stack order
  int fwd_emit = 1;         // LRP_FWD_EMIT=0      forward: split / absmax / pool passes between the convs
  int fwd_il = 1;           // LRP_FWD_IL=0        dual forward matrix with stacked rows: separate gate pass (decided when lrp_set_weight packs)
  int img_fused = 1;        // LRP_IMG_FUSED=0     image layer as T GEMM + separate stencil kernel (VGG16 and the ResNet stem)
  int up2_compact = 1;      // LRP_UP2_COMPACT=0   expanded pool interface between block2_conv1 and block1_conv2
  int up2_gc = 1;           // LRP_UP2_GC=0        that interface with the full-resolution pool gate
  int up2_breg_pairs = 1;   // LRP_UP2_BREG_PAIRS=0  block2_conv1 writes its plain fp32 product instead of pairs
  int img_fold = 1;         // LRP_IMG_FOLD=0      image layer as its own launch
  int dec_batched = 1;      // LRP_DEC_BATCHED=0   decoder LRP: one workgroup per unit instead of the step-synchronous scan
  int dec_mfma_fwd = 1;     // LRP_DEC_MFMA_FWD=0  decoder forward: VALU skinny GEMMs
  int pool_fused = 1;       // LRP_POOL_FUSED=0    forward: max-pool + gate + pooled pairs as a pass of their own behind the conv
  int sparse_pool = 0;      // LRP_SPARSE_POOL=1   pooled boundaries with >= 256 output columns on the 2:4-sparse matrix cores (conv_sparse.h); read by encode and explain
  void load() {
    *this = Switches();
    auto rd = [](const char* name, int& v) { if (const char* e = getenv(name)) v = atoi(e); };
    rd("LRP_CONV_HALO", conv_halo); rd("LRP_CONV_BREG", conv_breg); rd("LRP_CONV_TILE", conv_tile); rd("LRP_CONV_SMALL", conv_small);
    rd("LRP_CONV_MID", conv_mid); rd("LRP_EPI_FAST", epi_fast); rd("LRP_UP2_PW", up2_pw); rd("LRP_TILE_ORDER", tile_order);
    rd("LRP_FWD_EMIT", fwd_emit); rd("LRP_FWD_IL", fwd_il); rd("LRP_IMG_FUSED", img_fused); rd("LRP_UP2_COMPACT", up2_compact);
    rd("LRP_UP2_GC", up2_gc); rd("LRP_UP2_BREG_PAIRS", up2_breg_pairs); rd("LRP_IMG_FOLD", img_fold); rd("LRP_DEC_BATCHED", dec_batched);
    rd("LRP_DEC_MFMA_FWD", dec_mfma_fwd); rd("LRP_SPARSE_POOL", sparse_pool); rd("LRP_POOL_FUSED", pool_fused);
  }
};
inline Switches& sw() {
  static Switches s = [] { Switches t; t.load(); return t; }();
  return s;
}

inline int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  last_error_ref() = buf;
  return code;
}

#define LRP_HIP_CHECK(expr)                                                                     \
  do {                                                                                          \
    hipError_t _e = (expr);                                                                     \
    if (_e != hipSuccess)                                                                       \
      return ::lrp::fail(LRP_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                         __FILE__, __LINE__);                                                   \
  } while (0)

#define LRP_TRY(expr)         \
  do {                        \
    int _rc = (expr);         \
    if (_rc != LRP_OK) return _rc; \
  } while (0)

// Device allocation owned by the handle (workspace); freed in the destructor.
struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  bool owned = true;                                   // false: a window of another DevBuf (view_of), never freed here
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes), owned(o.owned) { o.p = nullptr; o.bytes = 0; o.owned = true; }
  DevBuf& operator=(DevBuf&& o) noexcept {
    if (this != &o) { release(); p = o.p; bytes = o.bytes; owned = o.owned; o.p = nullptr; o.bytes = 0; o.owned = true; }
    return *this;
  }
  ~DevBuf() { release(); }
  void release() {
    if (p && owned) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
    owned = true;
  }
  // n bytes at `offset` of `arena` (which must outlive this view)
  void view_of(const DevBuf& arena, size_t offset, size_t n) {
    release();
    p = static_cast<char*>(arena.p) + offset;
    bytes = n;
    owned = false;
  }
  int alloc(size_t n, int64_t* total) {
    release();
    if (n == 0) n = 16;
    hipError_t e = hipMalloc(&p, n);
    if (e != hipSuccess) {
      p = nullptr;
      (void)hipGetLastError();                         // clear the sticky error: the next launch check must not report this one
      return fail(LRP_ERR_NOMEM, "hipMalloc(%zu bytes) failed: %s", n, hipGetErrorString(e));
    }
    bytes = n;
    if (total) *total += (int64_t)n;
    return LRP_OK;
  }
  template <typename T>
  T* as() const { return reinterpret_cast<T*>(p); }
};

}  // namespace lrp
