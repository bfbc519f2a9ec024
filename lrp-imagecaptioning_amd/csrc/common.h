// common.h — error plumbing and small RAII helpers shared by the engine sources.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdarg>
#include <cstdio>
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
