// Shared helpers for libtrace_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <atomic>

#include "../../include/trace_amd.h"

namespace trace {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return 2;
  }
  return 0;
}

#define TRACE_REQUIRE(cond, ...)      \
  do {                                \
    if (!(cond)) {                    \
      ::trace::set_error(__VA_ARGS__); \
      return 1;                       \
    }                                 \
  } while (0)

// More than 64 KiB of dynamic LDS must be opted into once per kernel AND per device (a function attribute belongs
// to the current device's code object): `done` is the call site's per-device bitmap.
inline int opt_in_lds(std::atomic<uint64_t>& done, const void* fn, int bytes, const char* what) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) {
    set_error("%s: hipGetDevice: %s", what, hipGetErrorString(e));
    return 1;
  }
  const uint64_t bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_relaxed) & bit) return 0;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) {
    set_error("%s: hipFuncSetAttribute(%d bytes of LDS): %s", what, bytes, hipGetErrorString(e));
    return 1;
  }
  done.fetch_or(bit, std::memory_order_relaxed);
  return 0;
}
#define TRACE_LDS_OPT_IN(fn, bytes, what)                                                        \
  do {                                                                                           \
    static std::atomic<uint64_t> lds_done_{0};                                                   \
    if (::trace::opt_in_lds(lds_done_, reinterpret_cast<const void*>(fn), (int)(bytes), what)) return 1; \
  } while (0)

static inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

// bf16 <-> f32 (round-to-nearest-even through the hardware cvt where possible)
__device__ __forceinline__ float bf16_to_f32(uint16_t v) {
  return __uint_as_float(((uint32_t)v) << 16);
}
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
  // plain cast path: hipcc emits v_cvt_pk_bf16_f32 on gfx950 (keeps NaN a NaN)
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(uint16_t, b);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
}

}  // namespace trace
