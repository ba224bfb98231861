// TEST INFRASTRUCTURE ONLY (oracle/): forced-include used when compiling the
// reference's own CPU sources (Detectron_pytorch/lib/model/csrc/{vision.cpp,
// cpu/ROIAlign_cpu.cpp,cpu/nms_cpu.cpp}) where they lie under /root/reference.
// The sources call AT_DISPATCH_FLOATING_TYPES(tensor.type(), ...) which torch
// 2.10 only accepts with a c10::ScalarType; this header re-expresses that one
// macro so the unmodified sources compile.  It changes no arithmetic.
#pragma once
#include <torch/extension.h>
#include <ATen/Dispatch.h>
namespace trace_ref_compat {
inline c10::ScalarType st(const at::DeprecatedTypeProperties& t) { return t.scalarType(); }
inline c10::ScalarType st(c10::ScalarType t) { return t; }
}  // namespace trace_ref_compat
#undef AT_DISPATCH_FLOATING_TYPES
#define AT_DISPATCH_FLOATING_TYPES(TYPE, NAME, ...) \
  AT_DISPATCH_SWITCH(trace_ref_compat::st(TYPE), NAME, AT_DISPATCH_CASE_FLOATING_TYPES(__VA_ARGS__))
