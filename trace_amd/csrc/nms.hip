// Greedy NMS on device for gfx950.
//
// Reference behaviour: Detectron_pytorch/lib/utils/cython_nms.pyx:37-87 (the NMS
// TRACE runs at test time: ">= thresh", "+1" areas, ascending-index result),
// csrc/cpu/nms_cpu.cpp:5-65 (same algorithm), csrc/cuda/nms.cu:13-131 (64x64
// IoU tiles -> 64-bit masks, "> thresh", host-side reduce).
//
// MI355X design: the 64x64 tile of the CUDA kernel maps 1:1 onto one wave64
// (lane = row box, 64-bit suppression word per lane).  The reference's D2H copy
// + serial host loop is replaced by an on-device reduce: one wave walks the
// sorted boxes block by block, resolves the 64 boxes of the diagonal tile with
// register-only readlane steps, then ORs the kept rows' mask words (one word
// column per lane, coalesced) into the running "removed" bitmap.  Everything
// stays on the caller's stream; no host synchronisation.
//
// Compiled with -ffp-contract=off: IoU is the reference's f32 expression, bit
// for bit.
#include <mutex>

#include "common.h"

namespace trace {

constexpr int NMS_MAX_BOXES = 16384;  // 128 KiB of LDS for the sort; 256 mask words per row (4 per lane)

// trace_nms_noalloc: its scratch lives in the library's own device image (one copy per GPU, held until the process
// ends) instead of a caller workspace.  Layout = carve(NMS_STATIC_MAX_BOXES): order 16 KiB + sorted boxes 64 KiB +
// areas 16 KiB + suppression mask 4096 x 64 words = 2 MiB + keep flags 16 KiB + groups 16 KiB.
constexpr int NMS_STATIC_MAX_BOXES = 4096;
constexpr size_t NMS_STATIC_BYTES = 16384 + 65536 + 16384 + (size_t)4096 * 64 * 8 + 16384 + 16384;
static __device__ __align__(256) unsigned char scratch[NMS_STATIC_BYTES];

struct NmsWorkspace {
  int32_t* order;   // [Npad64] sorted -> original index
  float4* sboxes;   // [Npad64] boxes in sorted order
  float* sareas;    // [Npad64]
  uint64_t* mask;   // [Npad64 * col_blocks]
  int32_t* keep;    // [Npad64] flag per ORIGINAL index
  int32_t* sgroup;  // [Npad64] group id in sorted order (grouped NMS)
};

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static size_t carve(int N, void* base, NmsWorkspace* ws) {
  size_t npad = align_up((size_t)(N > 0 ? N : 1), 64);
  size_t col_blocks = npad / 64;
  size_t off = 0;
  char* p = (char*)base;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off += align_up(bytes, 256);
    return p ? (void*)(p + o) : (void*)nullptr;
  };
  void* order = take(npad * sizeof(int32_t));
  void* sboxes = take(npad * sizeof(float4));
  void* sareas = take(npad * sizeof(float));
  void* mask = take(npad * col_blocks * sizeof(uint64_t));
  void* keep = take(npad * sizeof(int32_t));
  void* sgroup = take(npad * sizeof(int32_t));
  if (ws) {
    ws->sgroup = (int32_t*)sgroup;
    ws->order = (int32_t*)order;
    ws->sboxes = (float4*)sboxes;
    ws->sareas = (float*)sareas;
    ws->mask = (uint64_t*)mask;
    ws->keep = (int32_t*)keep;
  }
  return off;
}

// "a sorts before b": descending score, ties by ascending original index.
__device__ __forceinline__ bool before(float sa, int ia, float sb, int ib) {
  return (sa > sb) || (sa == sb && ia < ib);
}

// Single-workgroup bitonic sort of (score, index) in LDS, then gather boxes.
__global__ __launch_bounds__(1024) void nms_sort_kernel(const float* __restrict__ boxes,
                                                        const float* __restrict__ scores, int N,
                                                        int npow2, int npad, int32_t* order,
                                                        float4* sboxes, float* sareas,
                                                        const int32_t* __restrict__ group,
                                                        int32_t* sgroup) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* key = (float*)smem;
  int* idx = (int*)(smem + (size_t)npow2 * sizeof(float));
  for (int i = threadIdx.x; i < npow2; i += blockDim.x) {
    if (i < N) {
      key[i] = scores[i];
      idx[i] = i;
    } else {
      key[i] = -__builtin_inff();
      idx[i] = 0x7fffffff;
    }
  }
  __syncthreads();
  for (int k = 2; k <= npow2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < npow2; i += blockDim.x) {
        int ixj = i ^ j;
        if (ixj > i) {
          float si = key[i], sj = key[ixj];
          int ii = idx[i], ij = idx[ixj];
          bool up = ((i & k) == 0);  // this run must end "sorted ascending position = before"
          bool swap = up ? before(sj, ij, si, ii) : before(si, ii, sj, ij);
          if (swap) {
            key[i] = sj;
            key[ixj] = si;
            idx[i] = ij;
            idx[ixj] = ii;
          }
        }
      }
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < npad; i += blockDim.x) {
    if (i < N) {
      int o = idx[i];
      order[i] = o;
      float x1 = boxes[o * 4 + 0], y1 = boxes[o * 4 + 1], x2 = boxes[o * 4 + 2],
            y2 = boxes[o * 4 + 3];
      sboxes[i] = make_float4(x1, y1, x2, y2);
      sareas[i] = (x2 - x1 + 1.0f) * (y2 - y1 + 1.0f);  // cython_nms.pyx:44
      sgroup[i] = group ? group[o] : 0;
    } else {
      sgroup[i] = -1;
      order[i] = -1;
      sboxes[i] = make_float4(0.f, 0.f, -1.f, -1.f);
      sareas[i] = 0.f;
    }
  }
}

// One wave per (row block, col block >= row block): lane = row box.
__global__ __launch_bounds__(64) void nms_mask_kernel(const float4* __restrict__ sboxes,
                                                      const float* __restrict__ sareas, int N,
                                                      int col_blocks, float thresh, int mode,
                                                      const int32_t* __restrict__ sgroup,
                                                      uint64_t* __restrict__ mask) {
  const int row_blk = blockIdx.y, col_blk = blockIdx.x;
  if (col_blk < row_blk) return;
  __shared__ float4 cb[64];
  __shared__ float ca[64];
  __shared__ int cg[64];
  const int lane = threadIdx.x;
  cb[lane] = sboxes[col_blk * 64 + lane];
  ca[lane] = sareas[col_blk * 64 + lane];
  cg[lane] = sgroup[col_blk * 64 + lane];
  __syncthreads();
  const int row = row_blk * 64 + lane;
  const float4 a = sboxes[row];
  const float aa = sareas[row];
  const int ag = sgroup[row];
  const int col_size = min(N - col_blk * 64, 64);
  uint64_t t = 0;
  const int start = (row_blk == col_blk) ? lane + 1 : 0;
  if (row < N) {
    for (int i = start; i < col_size; ++i) {
      const float4 b = cb[i];
      // cython_nms.pyx:76-83
      float xx1 = fmaxf(a.x, b.x), yy1 = fmaxf(a.y, b.y);
      float xx2 = fminf(a.z, b.z), yy2 = fminf(a.w, b.w);
      float w = fmaxf(0.0f, xx2 - xx1 + 1.0f);
      float h = fmaxf(0.0f, yy2 - yy1 + 1.0f);
      float inter = w * h;
      float ovr = inter / (aa + ca[i] - inter);
      bool sup = (mode == TRACE_NMS_GT) ? (ovr > thresh) : (ovr >= thresh);
      if (sup && cg[i] == ag) t |= 1ULL << i;  // boxes of different groups (classes) never interact
    }
  }
  mask[(size_t)row * col_blocks + col_blk] = t;
}

__device__ __forceinline__ uint64_t readlane64(uint64_t v, int lane) {
  uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, lane);
  uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(v >> 32), lane);
  return ((uint64_t)hi << 32) | lo;
}

// One wave: sequential greedy pass over the sorted boxes (nms.cu:105-123 moved
// on device).  Lane l owns removed-bitmap words l, 64+l, 128+l, 192+l.
__global__ __launch_bounds__(64) void nms_reduce_kernel(const uint64_t* __restrict__ mask,
                                                        const int32_t* __restrict__ order, int N,
                                                        int col_blocks, int32_t* __restrict__ keep) {
  const int lane = threadIdx.x;
  uint64_t remv0 = 0, remv1 = 0, remv2 = 0, remv3 = 0;
  for (int b = 0; b < col_blocks; ++b) {
    const int row = b * 64 + lane;
    const uint64_t diag = mask[(size_t)row * col_blocks + b];  // rows >= N hold 0
    const int slot = b >> 6, owner = b & 63;
    uint64_t mine = slot == 0 ? remv0 : slot == 1 ? remv1 : slot == 2 ? remv2 : remv3;
    uint64_t rb = readlane64(mine, owner);
    const int rows_here = min(N - b * 64, 64);
    uint64_t kept = 0;
    for (int r = 0; r < rows_here; ++r) {
      if (!((rb >> r) & 1ULL)) {
        kept |= 1ULL << r;
        rb |= readlane64(diag, r);
      }
    }
    if (lane < rows_here) keep[order[row]] = (int)((kept >> lane) & 1ULL);
    // OR the kept rows into the bitmap: every row of the block is loaded
    // unconditionally (8 in flight) and masked, so the loads pipeline.
    const size_t base = (size_t)b * 64 * col_blocks;
    for (int r0 = 0; r0 < 64; r0 += 8) {
      uint64_t v[8][4];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint64_t* rowp = mask + base + (size_t)(r0 + j) * col_blocks;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int w = k * 64 + lane;
          v[j][k] = (w > b && w < col_blocks) ? rowp[w] : 0ULL;
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint64_t m = ((kept >> (r0 + j)) & 1ULL) ? ~0ULL : 0ULL;
        remv0 |= v[j][0] & m;
        remv1 |= v[j][1] & m;
        remv2 |= v[j][2] & m;
        remv3 |= v[j][3] & m;
      }
    }
  }
}

// keep flags (by original index) -> ascending index list + count.
__global__ __launch_bounds__(1024) void nms_compact_kernel(const int32_t* __restrict__ keep, int N,
                                                           int64_t* __restrict__ keep_out,
                                                           int32_t* __restrict__ num_keep) {
  __shared__ int part[1024];
  const int t = threadIdx.x;
  const int per = (N + 1023) / 1024;
  const int lo = min(t * per, N), hi = min(lo + per, N);
  int cnt = 0;
  for (int i = lo; i < hi; ++i) cnt += keep[i] != 0;
  part[t] = cnt;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    int v = (t >= off) ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int pos = part[t] - cnt;
  for (int i = lo; i < hi; ++i)
    if (keep[i] != 0) keep_out[pos++] = i;
  if (t == 1023) *num_keep = part[1023];
}

}  // namespace trace

using namespace trace;

extern "C" size_t trace_nms_workspace_bytes(int n_boxes) {
  if (n_boxes < 0) n_boxes = 0;
  return carve(n_boxes, nullptr, nullptr);
}

static int nms_launch(const float* boxes, const float* scores, const int32_t* group, int N, float thresh,
                      int mode, int64_t* keep_out, int32_t* num_keep, void* workspace,
                      size_t workspace_bytes, trace_stream_t stream) {
  TRACE_REQUIRE(N >= 0, "nms: negative box count %d", N);
  TRACE_REQUIRE(N <= NMS_MAX_BOXES, "nms: %d boxes exceeds the supported maximum %d", N,
                NMS_MAX_BOXES);
  TRACE_REQUIRE(mode == TRACE_NMS_GE || mode == TRACE_NMS_GT, "nms: unknown mode %d", mode);
  TRACE_REQUIRE(num_keep, "nms: null num_keep");
  hipStream_t s = (hipStream_t)stream;
  if (N == 0) {  // csrc/nms.h:17-18 — empty in, empty out
    hipError_t e = hipMemsetAsync(num_keep, 0, sizeof(int32_t), s);
    TRACE_REQUIRE(e == hipSuccess, "nms: memset: %s", hipGetErrorString(e));
    return 0;
  }
  TRACE_REQUIRE(boxes && scores && keep_out && workspace, "nms: null pointer");
  NmsWorkspace ws;
  size_t need = carve(N, workspace, &ws);
  TRACE_REQUIRE(workspace_bytes >= need, "nms: workspace too small (%zu < %zu)", workspace_bytes,
                need);
  TRACE_REQUIRE(((uintptr_t)workspace & 255) == 0, "nms: workspace must be 256-byte aligned");
  int npad = (N + 63) / 64 * 64;
  int col_blocks = npad / 64;
  int npow2 = 64;
  while (npow2 < N) npow2 <<= 1;
  size_t lds = (size_t)npow2 * 8;
  if (lds > 64 * 1024) {
    TRACE_LDS_OPT_IN(&nms_sort_kernel, NMS_MAX_BOXES * 8, "nms");
  }
  hipLaunchKernelGGL(nms_sort_kernel, dim3(1), dim3(1024), lds, s, boxes, scores, N, npow2, npad,
                     ws.order, ws.sboxes, ws.sareas, group, ws.sgroup);
  hipLaunchKernelGGL(nms_mask_kernel, dim3(col_blocks, col_blocks), dim3(64), 0, s, ws.sboxes,
                     ws.sareas, N, col_blocks, thresh, mode, ws.sgroup, ws.mask);
  hipLaunchKernelGGL(nms_reduce_kernel, dim3(1), dim3(64), 0, s, ws.mask, ws.order, N, col_blocks,
                     ws.keep);
  hipLaunchKernelGGL(nms_compact_kernel, dim3(1), dim3(1024), 0, s, ws.keep, N, keep_out, num_keep);
  return check_launch("nms");
}

extern "C" int trace_nms(const float* boxes, const float* scores, int N, float thresh, int mode,
                         int64_t* keep_out, int32_t* num_keep, void* workspace,
                         size_t workspace_bytes, trace_stream_t stream) {
  return nms_launch(boxes, scores, nullptr, N, thresh, mode, keep_out, num_keep, workspace, workspace_bytes,
                    stream);
}

extern "C" int trace_nms_grouped(const float* boxes, const float* scores, const int32_t* group, int N,
                                 float thresh, int mode, int64_t* keep_out, int32_t* num_keep,
                                 void* workspace, size_t workspace_bytes, trace_stream_t stream) {
  TRACE_REQUIRE(N == 0 || group, "nms_grouped: null group array");
  return nms_launch(boxes, scores, group, N, thresh, mode, keep_out, num_keep, workspace, workspace_bytes,
                    stream);
}

// ---------------------------------------------------------------------------------------------------------
// The entry point that allocates nothing (see include/trace_amd.h).  `scratch` above is a host-side stand-in
// in host code, not a device address: the address of each GPU's own copy comes from hipGetSymbolAddress, once
// per device with that device current, and only that pointer is ever handed to a kernel.
// ---------------------------------------------------------------------------------------------------------
namespace trace {
constexpr int MAX_DEVICES = 64;
static void* g_scratch_ptr[MAX_DEVICES];
static std::mutex g_scratch_mu;

static int scratch_of_current_device(void** out) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  TRACE_REQUIRE(e == hipSuccess, "nms_noalloc: hipGetDevice: %s", hipGetErrorString(e));
  TRACE_REQUIRE(dev >= 0 && dev < MAX_DEVICES, "nms_noalloc: device index %d out of range", dev);
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  if (!g_scratch_ptr[dev]) {
    void* p = nullptr;
    e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    TRACE_REQUIRE(e == hipSuccess && p, "nms_noalloc: hipGetSymbolAddress: %s", hipGetErrorString(e));
    g_scratch_ptr[dev] = p;
  }
  *out = g_scratch_ptr[dev];
  return 0;
}
}  // namespace trace

extern "C" int trace_nms_noalloc(const float* boxes, const float* scores, int N, float thresh, int mode,
                                 int64_t* keep_out, int32_t* num_keep, trace_stream_t stream) {
  TRACE_REQUIRE(N <= NMS_STATIC_MAX_BOXES, "nms_noalloc: %d boxes exceeds the built-in scratch (max %d boxes); use "
                "trace_nms with a caller workspace", N, NMS_STATIC_MAX_BOXES);
  static_assert(NMS_STATIC_MAX_BOXES % 64 == 0, "scratch layout assumes whole 64-box blocks");
  void* ws = nullptr;
  if (scratch_of_current_device(&ws)) return 1;
  TRACE_REQUIRE(carve(NMS_STATIC_MAX_BOXES, nullptr, nullptr) <= NMS_STATIC_BYTES, "nms_noalloc: scratch layout outgrew its array");
  // every word the kernels read (sorted arrays, upper-triangle mask words, keep flags) is written by this call
  // first, so nothing of the previous call's contents has to be cleared
  return nms_launch(boxes, scores, nullptr, N, thresh, mode, keep_out, num_keep, ws, NMS_STATIC_BYTES, stream);
}
