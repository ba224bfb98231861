// Box helpers on device (bit-exact f32 restatements of the reference's numpy /
// Cython helpers).  Compiled with -ffp-contract=off.
//
//   fpn levels        Detectron_pytorch/lib/utils/fpn.py:11-28, utils/boxes.py:58-73
//   bbox_overlaps     Detectron_pytorch/lib/utils/cython_bbox.pyx:32-73
//   pair overlaps     lib/utils_rel/cython_bbox_rel.pyx:37-77
//   relpn pairs       lib/modeling_rel/relpn_heads.py:39-46,79-84,117;
//                     lib/utils_rel/boxes_rel.py:100-106
#include "common.h"

namespace trace {

__global__ __launch_bounds__(256) void fpn_levels_kernel(const float* __restrict__ boxes, int N,
                                                         float k_min, float k_max, float s0,
                                                         float lvl0, float* __restrict__ levels) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  float x1 = boxes[i * 4 + 0], y1 = boxes[i * 4 + 1], x2 = boxes[i * 4 + 2], y2 = boxes[i * 4 + 3];
  float w = x2 - x1 + 1.0f;           // boxes.py:60
  float h = y2 - y1 + 1.0f;           // boxes.py:61
  float area = w * h;                 // boxes.py:62
  if (area < 0.0f) area = 0.0f;       // fpn.py:17
  float s = sqrtf(area);              // fpn.py:18 (correctly rounded)
  float r = s / s0 + 1e-6f;           // fpn.py:23, f32 array (+) python float -> f32
  // log2 evaluated in f64 and rounded once to f32: the correctly rounded f32
  // log2, which is what numpy's f32 loop returns.
  float lg = (float)log2((double)r);
  float lvl = floorf(lvl0 + lg);      // fpn.py:23
  lvl = fminf(fmaxf(lvl, k_min), k_max);  // fpn.py:24
  levels[i] = lvl;
}

// IoU with "+ 1" extents exactly as the reference's Cython modules compute it: Cython emits the integer literal beside a
// float32 operand as the C double 1.0, so the coordinate differences are float32 while "+ 1", the area products and the
// sum of the union run in double and are rounded to float32 when they are assigned (box_area, iw, ih, ua are float32
// variables); iw * ih and the final division are float32.  Pinned bit for bit by tests/golden/cy_overlaps.npz (outputs
// of the compiled .pyx files); an all-float32 form is 1 ulp off in 1-5 % of the entries.
__device__ __forceinline__ float iou_plus1_cython(const float* b, const float* q) {
  const float box_area = (float)(((double)(q[2] - q[0]) + 1.0) * ((double)(q[3] - q[1]) + 1.0));
  const float iw = (float)((double)(fminf(b[2], q[2]) - fmaxf(b[0], q[0])) + 1.0);
  if (!(iw > 0.f)) return 0.f;
  const float ih = (float)((double)(fminf(b[3], q[3]) - fmaxf(b[1], q[1])) + 1.0);
  if (!(ih > 0.f)) return 0.f;
  const float ua = (float)(((double)(b[2] - b[0]) + 1.0) * ((double)(b[3] - b[1]) + 1.0) + (double)box_area - (double)(iw * ih));
  return iw * ih / ua;
}

__global__ __launch_bounds__(256) void bbox_overlaps_kernel(const float* __restrict__ boxes, int N,
                                                            const float* __restrict__ query, int K,
                                                            float* __restrict__ out) {
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)N * K) return;
  int n = (int)(idx / K), k = (int)(idx % K);
  const float* q = query + k * 4;
  const float* b = boxes + n * 4;
  // cython_bbox.pyx:52-72 as the compiled module evaluates it (iou_plus1_cython below)
  out[idx] = iou_plus1_cython(b, q);
}

__global__ __launch_bounds__(256) void bbox_pair_overlaps_kernel(const float* __restrict__ b1,
                                                                 const float* __restrict__ b2, int N,
                                                                 float* __restrict__ out) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float* a = b1 + n * 4;
  const float* q = b2 + n * 4;
  // cython_bbox_rel.pyx:57-76, same expression
  out[n] = iou_plus1_cython(a, q);
}

// All ordered pairs (s,o), s != o, in the reference's repeat/tile order
// (relpn_heads.py:79-84): pair p -> s = p / (n-1), o skips s.
__global__ __launch_bounds__(256) void relpn_pairs_kernel(const float* __restrict__ rois, int n,
                                                          int P, int32_t* __restrict__ sbj,
                                                          int32_t* __restrict__ obj,
                                                          float* __restrict__ rel) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  int s, o;
  if (n == 1) {
    s = 0;
    o = 0;
  } else {
    s = p / (n - 1);
    int r = p - s * (n - 1);
    o = r + (r >= s ? 1 : 0);
  }
  sbj[p] = s;
  obj[p] = o;
  const float* a = rois + s * 5;
  const float* b = rois + o * 5;
  rel[p * 5 + 0] = a[0];
  rel[p * 5 + 1] = fminf(a[1], b[1]);
  rel[p * 5 + 2] = fminf(a[2], b[2]);
  rel[p * 5 + 3] = fmaxf(a[3], b[3]);
  rel[p * 5 + 4] = fmaxf(a[4], b[4]);
}

// utils/boxes.py:160-200 bbox_transform followed by :142-157 clip_tiled_boxes: one thread per
// (roi, class) box.  deltas (R, 4*K); boxes (R,4); out (R, 4*K).
__global__ __launch_bounds__(256) void bbox_transform_clip_kernel(const float* __restrict__ boxes,
                                                                  const float* __restrict__ deltas, int R, int K,
                                                                  float wx, float wy, float ww, float wh,
                                                                  float xform_clip, float im_h, float im_w,
                                                                  float* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= R * K) return;
  const int r = idx / K;
  const float* b = boxes + (long long)r * 4;
  const float* d = deltas + (long long)idx * 4;
  const float width = b[2] - b[0] + 1.0f, height = b[3] - b[1] + 1.0f;
  const float ctr_x = b[0] + 0.5f * width, ctr_y = b[1] + 0.5f * height;
  const float dx = d[0] / wx, dy = d[1] / wy;
  const float dw = fminf(d[2] / ww, xform_clip), dh = fminf(d[3] / wh, xform_clip);
  const float pcx = dx * width + ctr_x, pcy = dy * height + ctr_y;
  const float pw = expf(dw) * width, ph = expf(dh) * height;
  float x1 = pcx - 0.5f * pw, y1 = pcy - 0.5f * ph;
  float x2 = pcx + 0.5f * pw - 1.0f, y2 = pcy + 0.5f * ph - 1.0f;
  x1 = fmaxf(fminf(x1, im_w - 1.0f), 0.f);
  y1 = fmaxf(fminf(y1, im_h - 1.0f), 0.f);
  x2 = fmaxf(fminf(x2, im_w - 1.0f), 0.f);
  y2 = fmaxf(fminf(y2, im_h - 1.0f), 0.f);
  *reinterpret_cast<float4*>(out + (long long)idx * 4) = make_float4(x1, y1, x2, y2);
}

}  // namespace trace

using namespace trace;

extern "C" int trace_fpn_levels(const float* boxes, int N, int k_min, int k_max,
                                float canonical_scale, float canonical_level, float* levels,
                                trace_stream_t stream) {
  TRACE_REQUIRE(N >= 0, "fpn_levels: negative count");
  if (N == 0) return 0;
  TRACE_REQUIRE(boxes && levels, "fpn_levels: null pointer");
  hipLaunchKernelGGL(fpn_levels_kernel, dim3(ceil_div(N, 256)), dim3(256), 0, (hipStream_t)stream,
                     boxes, N, (float)k_min, (float)k_max, canonical_scale, canonical_level, levels);
  return check_launch("fpn_levels");
}

extern "C" int trace_bbox_overlaps(const float* boxes, int N, const float* query, int K,
                                   float* overlaps, trace_stream_t stream) {
  TRACE_REQUIRE(N >= 0 && K >= 0, "bbox_overlaps: negative count");
  if ((long long)N * K == 0) return 0;
  TRACE_REQUIRE(boxes && query && overlaps, "bbox_overlaps: null pointer");
  hipLaunchKernelGGL(bbox_overlaps_kernel, dim3(ceil_div((long long)N * K, 256)), dim3(256), 0,
                     (hipStream_t)stream, boxes, N, query, K, overlaps);
  return check_launch("bbox_overlaps");
}

extern "C" int trace_bbox_pair_overlaps(const float* boxes1, const float* boxes2, int N,
                                        float* overlaps, trace_stream_t stream) {
  TRACE_REQUIRE(N >= 0, "bbox_pair_overlaps: negative count");
  if (N == 0) return 0;
  TRACE_REQUIRE(boxes1 && boxes2 && overlaps, "bbox_pair_overlaps: null pointer");
  hipLaunchKernelGGL(bbox_pair_overlaps_kernel, dim3(ceil_div(N, 256)), dim3(256), 0,
                     (hipStream_t)stream, boxes1, boxes2, N, overlaps);
  return check_launch("bbox_pair_overlaps");
}

extern "C" int trace_relpn_pairs(const float* det_rois, int n, int32_t* sbj_inds, int32_t* obj_inds,
                                 float* rel_rois, trace_stream_t stream) {
  TRACE_REQUIRE(n >= 0, "relpn_pairs: negative count");
  if (n == 0) return 0;
  TRACE_REQUIRE(n <= 46340, "relpn_pairs: n=%d overflows the pair index", n);
  int P = (n == 1) ? 1 : n * (n - 1);
  TRACE_REQUIRE(det_rois && sbj_inds && obj_inds && rel_rois, "relpn_pairs: null pointer");
  hipLaunchKernelGGL(relpn_pairs_kernel, dim3(ceil_div(P, 256)), dim3(256), 0, (hipStream_t)stream,
                     det_rois, n, P, sbj_inds, obj_inds, rel_rois);
  return check_launch("relpn_pairs");
}

extern "C" int trace_bbox_transform_clip(const float* boxes, const float* deltas, int R, int K, float wx,
                                         float wy, float ww, float wh, float xform_clip, float im_h,
                                         float im_w, float* out, trace_stream_t stream) {
  TRACE_REQUIRE(R >= 0 && K > 0, "bbox_transform_clip: bad shape R=%d K=%d", R, K);
  if (R == 0) return 0;
  TRACE_REQUIRE(boxes && deltas && out, "bbox_transform_clip: null pointer");
  TRACE_REQUIRE(((uintptr_t)out & 15) == 0, "bbox_transform_clip: out must be 16-byte aligned");
  const long long n = (long long)R * K;
  hipLaunchKernelGGL(bbox_transform_clip_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, boxes, deltas, R, K, wx, wy, ww, wh, xform_clip, im_h, im_w, out);
  return check_launch("bbox_transform_clip");
}
