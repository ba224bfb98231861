// Shared helpers for libtrace_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

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
