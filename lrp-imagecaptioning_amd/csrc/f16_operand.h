// f16_operand.h — fp16-pair copies of packed fp32 weight matrices, made on the device (shared by the VGG16 and the
// ResNet-101 encoder handles).
#pragma once
#include <hip/hip_runtime.h>
#include "cnn_kernels.h"
#include "common.h"

namespace lrp {

// max|w| -> power-of-two scale -> pairs [hi8|lo8] of (w * scale) -> (rows > 0) the norm of the scaled matrix; the record
// {2^k, 2^-k, norm, k} lands in `wsc` (cnn_kernels.h: wscale_kernel).  Without the scale the low halves of small weights
// fall into fp16 subnormals.  `slots`: ACT_MAX_SLOTS scratch maxima, allocated on first use.  sync = true for host
// setters (their source upload was synchronous, so is this); device packers pass their stream and sync = false.
inline int make_f16_operand(DevBuf& slots, const float* src, size_t n_floats, int rows, int K, DevBuf& dst, DevBuf& wsc,
                            int64_t* total, hipStream_t st, bool sync = true) {
  if (!slots.p) LRP_TRY(slots.alloc(ACT_MAX_SLOTS * sizeof(unsigned), total));
  if (!wsc.p) LRP_TRY(wsc.alloc(4 * sizeof(float), total));
  if (!dst.p || dst.bytes != n_floats * sizeof(float)) LRP_TRY(dst.alloc(n_floats * sizeof(float), total));
  LRP_HIP_CHECK(hipMemsetAsync(slots.p, 0, ACT_MAX_SLOTS * sizeof(unsigned), st));
  hipLaunchKernelGGL(absmax_slots_kernel, dim3(stream_grid(n_floats / 4)), dim3(256), 0, st, reinterpret_cast<const f32x4*>(src),
                     n_floats / 4, slots.as<unsigned>());
  hipLaunchKernelGGL(wscale_kernel, dim3(1), dim3(64), 0, st, slots.as<unsigned>(), wsc.as<float>());
  hipLaunchKernelGGL(split_copy_h_kernel, dim3(stream_grid(n_floats / 8)), dim3(256), 0, st, src, dst.as<float>(), n_floats / 8,
                     wsc.as<float>());
  if (rows > 0) hipLaunchKernelGGL(rowabs_max_kernel, dim3(rows), dim3(256), 0, st, src, K, wsc.as<float>());
  LRP_HIP_CHECK(hipGetLastError());
  if (sync) LRP_HIP_CHECK(hipStreamSynchronize(st));
  return LRP_OK;
}

}  // namespace lrp
