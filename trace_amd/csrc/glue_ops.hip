// Elementwise / gather glue of the TRACE relation head (gfx950), one launch per chain of torch ops that
// the reference issues between its Linear layers.  All HBM-bound, tiny (<= a few MB per step); they exist
// because inside the captured step every extra launch costs ~4 us of dependent-kernel boundary.
//
//   act_cast          F.relu / LeakyReLU(0.1) feeding a Linear (e.g. lib/modeling_rel/model_builder_rel.py:273,
//                     281-286, fast_rcnn_heads.py:45 via :939), and torch.cat of the tree-GRU's h | hd
//                     (lib/combine_tree/cluster_tree.py:203)
//   layernorm_act     nn.LayerNorm + ReLU of the temporal value MLPs (lib/combine_tree/cluster_tree3d.py:80-84,91-95)
//   gather_add_act    sem_feat_vec = low_s[X] + low_r[lca] + low_o[Y], then LeakyReLU (model_builder_rel.py:1662, :326)
//   pair_assemble     per-pair gathers of labels / scores / subject+object terms and the frequency bias
//                     (model_builder_rel.py:951-952 [gathers], :1064-1069, lib/modeling_rel/sparse_targets_rel.py:67-72,
//                     lib/modeling_rel/relpn_heads.py:113-116)
#include "common.h"

namespace trace {

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act & TRACE_EPI_RELU) v = fmaxf(v, 0.f);
  if (act & TRACE_EPI_LEAKY) v = v > 0.f ? v : 0.1f * v;
  return v;
}

// out (rows, wa+wb) bf16 = act([a | b]); one thread per 4 output columns (wa, wb multiples of 4).
__global__ __launch_bounds__(256) void act_cast_kernel(const float* __restrict__ a, int wa, const float* __restrict__ b,
                                                       int wb, long long rows, int act, uint16_t* __restrict__ out) {
  const int wq = (wa + wb) >> 2;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * wq) return;
  const long long r = idx / wq;
  const int c = (int)(idx % wq) * 4;
  const float4 v = c < wa ? *reinterpret_cast<const float4*>(a + r * wa + c)
                          : *reinterpret_cast<const float4*>(b + r * wb + (c - wa));
  uint2 o;
  o.x = pack_bf16x2(apply_act(v.x, act), apply_act(v.y, act));
  o.y = pack_bf16x2(apply_act(v.z, act), apply_act(v.w, act));
  *reinterpret_cast<uint2*>(out + r * (wa + wb) + c) = o;
}

// One wave per row: y = act((x - mean) / sqrt(var + eps) * gamma + beta) -> bf16 (biased variance, as torch).
__global__ __launch_bounds__(256) void layernorm_act_kernel(const float* __restrict__ x, int rows, int w,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps, int act,
                                                            uint16_t* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* xr = x + (long long)r * w;
  float s = 0.f;
  for (int j = lane; j < w; j += 64) s += xr[j];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  const float mean = s / (float)w;
  float q = 0.f;
  for (int j = lane; j < w; j += 64) {
    const float d = xr[j] - mean;
    q += d * d;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off);
  const float rstd = rsqrtf(q / (float)w + eps);
  for (int j = lane; j < w; j += 64) {
    float v = (xr[j] - mean) * rstd;
    v = v * (gamma ? gamma[j] : 1.0f) + (beta ? beta[j] : 0.f);
    out[(long long)r * w + j] = f32_to_bf16(apply_act(v, act));
  }
}

// out (rows, w) bf16 = act(a[r] + b[idx[r]]); w multiple of 4.
__global__ __launch_bounds__(256) void gather_add_act_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                             const int64_t* __restrict__ idx, long long rows, int w,
                                                             int act, uint16_t* __restrict__ out) {
  const int wq = w >> 2;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * wq) return;
  const long long r = i / wq;
  const int c = (int)(i % wq) * 4;
  const float4 va = *reinterpret_cast<const float4*>(a + r * w + c);
  const float4 vb = *reinterpret_cast<const float4*>(b + idx[r] * w + c);
  uint2 o;
  o.x = pack_bf16x2(apply_act(va.x + vb.x, act), apply_act(va.y + vb.y, act));
  o.y = pack_bf16x2(apply_act(va.z + vb.z, act), apply_act(va.w + vb.w, act));
  *reinterpret_cast<uint2*>(out + r * w + c) = o;
}

// One workgroup per pair p (subject X[p], object Y[p], indices into the step's detection list).
__global__ __launch_bounds__(128) void pair_assemble_kernel(
    const int64_t* __restrict__ det_labels, const float* __restrict__ det_scores, const int64_t* __restrict__ X,
    const int64_t* __restrict__ Y, const int64_t* __restrict__ pls, const int64_t* __restrict__ plo,
    const float* __restrict__ pss, const float* __restrict__ pso, const float* __restrict__ low_s,
    const float* __restrict__ low_o, int wl,
    const float* __restrict__ part_s, const float* __restrict__ part_o, int wp, const float* __restrict__ freq,
    int n_rel, int n_obj, int64_t* __restrict__ sbj_lab, int64_t* __restrict__ obj_lab, float* __restrict__ sbj_sc,
    float* __restrict__ obj_sc, float* __restrict__ so_bias, float* __restrict__ so_part) {
  const long long p = blockIdx.x;
  const long long s = X[p], o = Y[p];
  // per-pair labels / scores when given (use_gt_labels, model_builder_rel.py:844-858), else gathered per roi
  const int64_t ls = pls ? pls[p] : det_labels[s], lo = plo ? plo[p] : det_labels[o];
  if (threadIdx.x == 0) {
    sbj_lab[p] = ls;   // relpn_heads.py:113-116
    obj_lab[p] = lo;
    sbj_sc[p] = pss ? pss[p] : det_scores[s];
    obj_sc[p] = pso ? pso[p] : det_scores[o];
  }
  for (int j = threadIdx.x; j < wl; j += blockDim.x)  // model_builder_rel.py:1662 (the region term joins later)
    so_bias[p * wl + j] = low_s[s * wl + j] + low_o[o * wl + j];
  // sparse_targets_rel.py:67-72: rel_baseline[(sbj-1) * n_obj + (obj-1)] (labels are 1-based, :1064)
  const float* fb = freq ? freq + ((ls - 1) * n_obj + (lo - 1)) * n_rel : nullptr;
  for (int j = threadIdx.x; j < wp; j += blockDim.x) {
    float v = part_s[s * wp + j] + part_o[o * wp + j];  // :1066-1069 sem_s(sbj_feat) + sem_o(obj_feat)
    if (fb && j < n_rel) v += fb[j];                    // :1075
    so_part[p * wp + j] = v;
  }
}

}  // namespace trace

using namespace trace;

extern "C" int trace_act_cast_bf16(const float* a, int wa, const float* b, int wb, long long rows, int act,
                                   uint16_t* out, trace_stream_t stream) {
  TRACE_REQUIRE(rows >= 0 && wa > 0 && wb >= 0 && wa % 4 == 0 && wb % 4 == 0, "act_cast: widths must be multiples of 4");
  TRACE_REQUIRE((act & ~(TRACE_EPI_RELU | TRACE_EPI_LEAKY)) == 0, "act_cast: act takes RELU / LEAKY only");
  if (rows == 0) return 0;
  TRACE_REQUIRE(a && out && (wb == 0 || b), "act_cast: null pointer");
  TRACE_REQUIRE((((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) == 0, "act_cast: pointers must be 16-byte aligned");
  const long long n = rows * ((wa + wb) / 4);
  TRACE_REQUIRE(n < (1LL << 31) * 256, "act_cast: too large");
  hipLaunchKernelGGL(act_cast_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, wa, b, wb,
                     rows, act, out);
  return check_launch("act_cast_bf16");
}

extern "C" int trace_layernorm_act_bf16(const float* x, int rows, int w, const float* gamma, const float* beta,
                                        float eps, int act, uint16_t* out, trace_stream_t stream) {
  TRACE_REQUIRE(rows >= 0 && w > 0, "layernorm_act: bad shape");
  TRACE_REQUIRE((act & ~(TRACE_EPI_RELU | TRACE_EPI_LEAKY)) == 0, "layernorm_act: act takes RELU / LEAKY only");
  if (rows == 0) return 0;
  TRACE_REQUIRE(x && out, "layernorm_act: null pointer");
  hipLaunchKernelGGL(layernorm_act_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, rows, w, gamma,
                     beta, eps, act, out);
  return check_launch("layernorm_act_bf16");
}

extern "C" int trace_gather_add_act_bf16(const float* a, const float* b, const int64_t* idx, long long rows, int w,
                                         int act, uint16_t* out, trace_stream_t stream) {
  TRACE_REQUIRE(rows >= 0 && w > 0 && w % 4 == 0, "gather_add_act: width must be a multiple of 4");
  TRACE_REQUIRE((act & ~(TRACE_EPI_RELU | TRACE_EPI_LEAKY)) == 0, "gather_add_act: act takes RELU / LEAKY only");
  if (rows == 0) return 0;
  TRACE_REQUIRE(a && b && idx && out, "gather_add_act: null pointer");
  TRACE_REQUIRE((((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) == 0, "gather_add_act: pointers must be 16-byte aligned");
  const long long n = rows * (w / 4);
  hipLaunchKernelGGL(gather_add_act_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, b,
                     idx, rows, w, act, out);
  return check_launch("gather_add_act_bf16");
}

extern "C" int trace_pair_assemble_ex(const int64_t* det_labels, const float* det_scores, const int64_t* X,
                                      const int64_t* Y, int P, const int64_t* pair_sbj_lab, const int64_t* pair_obj_lab,
                                      const float* pair_sbj_sc, const float* pair_obj_sc, const float* low_s,
                                      const float* low_o, int wl, const float* part_s, const float* part_o, int wp,
                                      const float* freq, int n_rel, int n_obj, int64_t* sbj_lab, int64_t* obj_lab,
                                      float* sbj_sc, float* obj_sc, float* so_bias, float* so_part, trace_stream_t stream) {
  TRACE_REQUIRE(P >= 0 && wl > 0 && wp > 0 && n_rel >= 0 && n_rel <= wp && n_obj >= 0, "pair_assemble: bad shape");
  if (P == 0) return 0;
  TRACE_REQUIRE(X && Y && low_s && low_o && part_s && part_o && sbj_lab && obj_lab && sbj_sc && obj_sc && so_bias && so_part,
                "pair_assemble: null pointer");
  TRACE_REQUIRE((det_labels || (pair_sbj_lab && pair_obj_lab)) && (det_scores || (pair_sbj_sc && pair_obj_sc)),
                "pair_assemble: labels / scores must come per roi or per pair");
  hipLaunchKernelGGL(pair_assemble_kernel, dim3(P), dim3(128), 0, (hipStream_t)stream, det_labels, det_scores, X, Y,
                     pair_sbj_lab, pair_obj_lab, pair_sbj_sc, pair_obj_sc, low_s, low_o, wl, part_s, part_o, wp, freq, n_rel,
                     n_obj, sbj_lab, obj_lab, sbj_sc, obj_sc, so_bias, so_part);
  return check_launch("pair_assemble");
}

extern "C" int trace_pair_assemble(const int64_t* det_labels, const float* det_scores, const int64_t* X,
                                   const int64_t* Y, int P, const float* low_s, const float* low_o, int wl,
                                   const float* part_s, const float* part_o, int wp, const float* freq, int n_rel,
                                   int n_obj, int64_t* sbj_lab, int64_t* obj_lab, float* sbj_sc, float* obj_sc,
                                   float* so_bias, float* so_part, trace_stream_t stream) {
  TRACE_REQUIRE(P == 0 || (det_labels && det_scores), "pair_assemble: null pointer");
  return trace_pair_assemble_ex(det_labels, det_scores, X, Y, P, nullptr, nullptr, nullptr, nullptr, low_s, low_o, wl, part_s,
                                part_o, wp, freq, n_rel, n_obj, sbj_lab, obj_lab, sbj_sc, obj_sc, so_bias, so_part, stream);
}

// ---------------------------------------------------------------------------------------------
// Per-step host arrays without a copy-engine hop: the first node of a captured step reads the step's
// packed host arrays (roi lists, forest CSR arrays, pair indices - what the reference uploads with ~20
// blocking torch.from_numpy(..).cuda() per clip, lib/modeling_rel/model_builder_rel.py:1389,
// lib/combine_tree/cluster_tree.py:214,730) straight from pinned host memory.  The pinned buffer holds
// two slots; launch number k reads slot k & 1 (k lives in `counter`), so the host may fill the other
// slot while the GPU still runs the previous step.
namespace trace {
// counter[0] = launch number, counter[1] = workgroups of the current launch that have read it: the LAST one to
// arrive bumps the launch number (every workgroup has its copy of `seq` by then) and clears the arrival count.
__global__ __launch_bounds__(256) void upload_slots_kernel(const uint4* __restrict__ pinned, size_t slot_u4,
                                                           unsigned* counter, uint4* __restrict__ dst, size_t n_u4) {
  __shared__ unsigned s_seq;
  if (threadIdx.x == 0) {
    const unsigned seq = __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_seq = seq;
    __threadfence();   // the read of the launch number completes before this workgroup counts as arrived
    const unsigned arrived = atomicAdd(counter + 1, 1u);
    if (arrived == gridDim.x - 1) {
      atomicExch(counter + 1, 0u);
      atomicAdd(counter, 1u);
    }
  }
  __syncthreads();
  const uint4* src = pinned + (size_t)(s_seq & 1u) * slot_u4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_u4; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = src[i];
}
}  // namespace trace

extern "C" int trace_upload_slots(const void* pinned, size_t slot_bytes, unsigned* counter, void* dst, size_t nbytes,
                                  trace_stream_t stream) {
  TRACE_REQUIRE(pinned && counter && dst, "upload_slots: null pointer");
  TRACE_REQUIRE(slot_bytes % 16 == 0 && nbytes % 16 == 0 && nbytes <= slot_bytes, "upload_slots: sizes must be multiples of 16, nbytes <= slot_bytes");
  TRACE_REQUIRE((((uintptr_t)pinned | (uintptr_t)dst) & 15) == 0, "upload_slots: pointers must be 16-byte aligned");
  const size_t n_u4 = nbytes / 16;
  unsigned blocks = (unsigned)((n_u4 + 255) / 256);
  if (blocks > 32) blocks = 32;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(trace::upload_slots_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)pinned,
                     slot_bytes / 16, counter, (uint4*)dst, n_u4);
  return trace::check_launch("upload_slots");
}
