// Greedy NMS on device for gfx950.
//
// Reference behaviour: Detectron_pytorch/lib/utils/cython_nms.pyx:37-87 (the NMS
// TRACE runs at test time: ">= thresh", "+1" areas, ascending-index result),
// csrc/cpu/nms_cpu.cpp:5-65 (same algorithm), csrc/cuda/nms.cu:13-131 (64x64
// IoU tiles -> 64-bit masks, "> thresh", host-side reduce).
//
// MI355X design: the 64x64 tile of the CUDA kernel maps 1:1 onto one wave64
// (lane = row box, 64-bit suppression word per lane).  The sort is a rank sort
// spread over N/64 workgroups (integer keys, no barrier-separated network).  The
// reference's D2H copy + serial host loop is replaced by an on-device reduce:
// one workgroup walks the sorted boxes block by block; wave 0 resolves the 64
// boxes of the diagonal tile with one readlane step per KEPT box and carries the
// next two columns in registers, 15 helper waves OR the kept rows' remaining
// mask words into the "removed" bitmap in LDS two blocks behind, so no global
// load sits on the block-to-block critical path.  Everything stays on the
// caller's stream; no host synchronisation.
//
// Compiled with -ffp-contract=off: IoU is the reference's f32 expression, bit
// for bit.
#include <mutex>

#include "common.h"

namespace trace {

constexpr int NMS_MAX_BOXES = 16384;  // 256 mask words per row: 4 per lane in the reduce's helper waves

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

// Sort key: "a sorts before b" (descending score, ties by ascending original index) == key(a) > key(b).
// -0.0f sorts with +0.0f (the float comparison of the reference treats them as equal, then the index decides).
// Grouped NMS sorts by group first (ascending id, 18 bits: ids in [0, 2^18 - 2]), so that every group is one contiguous
// run of the sorted order and the groups can be reduced independently; group 0 everywhere gives the plain order.
// index < NMS_MAX_BOXES = 2^14.
__device__ __forceinline__ uint64_t sort_key(float score, int index, int group = 0) {
  uint32_t u = __float_as_uint(score + 0.0f);
  u ^= (u >> 31) ? 0xffffffffu : 0x80000000u;
  return ((uint64_t)(0x3ffffu - ((uint32_t)group & 0x3ffffu)) << 46) | ((uint64_t)u << 14) | (uint32_t)(0x3fff - index);
}

// Rank sort: position of box i in the sorted order = number of boxes that sort before it.  N^2 integer comparisons
// spread over N/64 workgroups (lane = box, wave = one sixteenth of a tile of the other boxes' keys, staged in LDS
// and read two per ds_read_b128 at a wave-uniform address) replace a single-workgroup bitonic network of 91
// barrier-separated stages; the ranks are a permutation, so the kernel scatters boxes, areas and groups to their
// sorted slots itself.
constexpr int NMS_SORT_TILE = 4096;   // keys per LDS tile (32 KiB)
__global__ __launch_bounds__(1024) void nms_sort_kernel(const float* __restrict__ boxes,
                                                        const float* __restrict__ scores, int N, int npad,
                                                        int32_t* __restrict__ order, float4* __restrict__ sboxes,
                                                        float* __restrict__ sareas,
                                                        const int32_t* __restrict__ group,
                                                        int32_t* __restrict__ sgroup) {
  __shared__ __attribute__((aligned(16))) unsigned long long keys[NMS_SORT_TILE];
  __shared__ int part[16][64];
  const int lane = threadIdx.x & 63, split = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  const uint64_t ki = sort_key(i < N ? scores[i] : 0.f, i < N ? i : 0, (group && i < N) ? group[i] : 0);
  int cnt = 0;
  for (int t0 = 0; t0 < N; t0 += NMS_SORT_TILE) {
    const int tn = min(NMS_SORT_TILE, N - t0);
    __syncthreads();
    // key 0 (below every real key: real keys have the index complement in the low word) pads the tile to 32 keys per wave
    for (int j = threadIdx.x; j < NMS_SORT_TILE; j += blockDim.x) keys[j] = j < tn ? sort_key(scores[t0 + j], t0 + j, group ? group[t0 + j] : 0) : 0ULL;
    __syncthreads();
    const int per = ((tn + 15) / 16 + 1) & ~1;   // keys per wave, even
    const int j0 = split * per, j1 = min(j0 + per, NMS_SORT_TILE);
    for (int j = j0; j < j1; j += 2) {
      const ulonglong2 k2 = *reinterpret_cast<const ulonglong2*>(&keys[j]);
      cnt += k2.x > ki;
      cnt += k2.y > ki;
    }
  }
  part[split][lane] = cnt;
  __syncthreads();
  if (split == 0 && i < N) {
    int r = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) r += part[k][lane];
    const float x1 = boxes[i * 4 + 0], y1 = boxes[i * 4 + 1], x2 = boxes[i * 4 + 2], y2 = boxes[i * 4 + 3];
    order[r] = i;
    sboxes[r] = make_float4(x1, y1, x2, y2);
    sareas[r] = (x2 - x1 + 1.0f) * (y2 - y1 + 1.0f);  // cython_nms.pyx:44
    sgroup[r] = group ? group[i] : 0;
  }
  if (blockIdx.x == 0) {
    for (int r = N + threadIdx.x; r < npad; r += blockDim.x) {
      sgroup[r] = -1;
      order[r] = -1;
      sboxes[r] = make_float4(0.f, 0.f, -1.f, -1.f);
      sareas[r] = 0.f;
    }
  }
}

// One wave per (row block, col block >= row block): lane = row box.
__global__ __launch_bounds__(64) void nms_mask_kernel(const float4* __restrict__ sboxes,
                                                      const float* __restrict__ sareas, int N,
                                                      int col_blocks, float thresh, int mode,
                                                      const int32_t* __restrict__ sgroup, int grouped,
                                                      uint64_t* __restrict__ mask) {
  const int row_blk = blockIdx.y, col_blk = blockIdx.x;
  if (col_blk < row_blk) return;
  // grouped: the boxes are sorted by group, so a column block whose first group lies behind the row block's last one
  // shares no group with it; nms_reduce_grouped_kernel never reads such a word
  if (grouped && sgroup[col_blk * 64] > sgroup[min(row_blk * 64 + 63, N - 1)]) return;
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
  // the diagonal tile holds BOTH triangles (the predicate is symmetric): bit i < lane of row `lane` says "earlier box
  // i suppresses me", which is what the reduce's parallel resolve reads
  const bool diag_tile = row_blk == col_blk;
  if (row < N) {
    for (int i = 0; i < col_size; ++i) {
      if (diag_tile && i == lane) continue;
      const float4 b = cb[i];
      // cython_nms.pyx:76-83
      float xx1 = fmaxf(a.x, b.x), yy1 = fmaxf(a.y, b.y);
      float xx2 = fminf(a.z, b.z), yy2 = fminf(a.w, b.w);
      float w = fmaxf(0.0f, xx2 - xx1 + 1.0f);
      float h = fmaxf(0.0f, yy2 - yy1 + 1.0f);
      float inter = w * h;
      // disjoint boxes: ovr is 0 / positive = +0 and the comparison below depends on thresh alone
      // (areas of degenerate boxes can make the union <= 0: those take the division like everything else)
      const float uni = aa + ca[i] - inter;
      float ovr = 0.f;
      if (inter != 0.f || !(uni > 0.f)) ovr = inter / uni;
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

// Greedy pass over the sorted boxes (nms.cu:105-123 moved on device), one workgroup of 16 waves.
// Box r of block b is kept iff no kept earlier box suppresses it, so block b needs word b of the "removed" bitmap
// complete - but nothing else.
//   Wave 0 walks the blocks.  The 64 boxes of the diagonal tile are resolved in parallel rounds (lane = box): a box
//   with a kept earlier suppressor is removed, a box whose earlier suppressors are all removed is kept; the lowest
//   undecided box always decides, and real tiles settle in a few rounds instead of 64 serial steps.  The kept lanes
//   then OR their words for the next THREE columns (prefetched one block ahead) into the bitmap in LDS themselves.
//   Waves 1..15 push the kept rows' remaining words (columns >= b + 4): loads issued in iteration b + 1 are OR-ed in
//   during iteration b + 3 (two register sets, alternating), so no global load sits on the block-to-block path.
constexpr int NMS_HELPERS = 15;
constexpr int NMS_ROWS_PER_HELPER = 5;   // ceil(64 / 15)
constexpr int NMS_CARRY = 3;             // columns wave 0 carries itself

template <int WPL>   // mask words per lane of a helper wave: 2 up to 128 + 4 column blocks, else 4
struct NmsPushSet {
  uint64_t v[NMS_ROWS_PER_HELPER][WPL];
  int base;   // first word column of v
};

template <int WPL>
__global__ __launch_bounds__(1024) void nms_reduce_kernel(const uint64_t* __restrict__ mask,
                                                          const int32_t* __restrict__ order, int N,
                                                          int col_blocks, int32_t* __restrict__ keep) {
  __shared__ unsigned long long remv[NMS_MAX_BOXES / 64 + NMS_CARRY];
  __shared__ int keptrow[2][64];   // kept rows of block it, compacted (slot it & 1); read by the helpers one iteration later
  __shared__ int keptcnt[2];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < NMS_MAX_BOXES / 64 + NMS_CARRY; i += blockDim.x) remv[i] = 0ULL;
  if (threadIdx.x < 2) keptcnt[threadIdx.x] = 0;
  __syncthreads();
  // wave 0: the next block's words and original indices, in flight while the current block resolves
  uint64_t diag = 0, nx[NMS_CARRY] = {0, 0, 0};
  int ord = -1;
  auto wave0_fetch = [&](int b) {   // rows >= N hold 0 in every word this kernel reads (nms_mask_kernel)
    const size_t row = (size_t)(b * 64 + lane) * col_blocks;
    diag = mask[row + b];
#pragma unroll
    for (int l = 0; l < NMS_CARRY; ++l) nx[l] = b + 1 + l < col_blocks ? mask[row + b + 1 + l] : 0ULL;
    ord = order[b * 64 + lane];
  };
  if (wv == 0) wave0_fetch(0);
  NmsPushSet<WPL> setA, setB;
#pragma unroll
  for (int j = 0; j < NMS_ROWS_PER_HELPER; ++j)
#pragma unroll
    for (int k = 0; k < WPL; ++k) setA.v[j][k] = setB.v[j][k] = 0ULL;
  setA.base = setB.base = 0;
  // helper step: OR in what this set loaded two iterations ago, then load the kept rows of block `blk`
  auto helper_step = [&](NmsPushSet<WPL>& st, int blk) {
    // the wave's (up to five) kept rows are OR-ed in registers first: one LDS atomic per word instead of one per row
#pragma unroll
    for (int k = 0; k < WPL; ++k) {
      uint64_t m = 0ULL;
#pragma unroll
      for (int j = 0; j < NMS_ROWS_PER_HELPER; ++j) {
        m |= st.v[j][k];
        st.v[j][k] = 0ULL;
      }
      if (m) atomicOr(&remv[st.base + k * 64 + lane], (unsigned long long)m);
    }
    if (blk < 0) return;
    const int cnt = keptcnt[blk & 1];
    st.base = blk + 1 + NMS_CARRY;
#pragma unroll
    for (int j = 0; j < NMS_ROWS_PER_HELPER; ++j) {
      const int want = (wv - 1) + NMS_HELPERS * j;   // this wave's j-th row: the kept row of that rank
      if (want < cnt) {
        const int r = keptrow[blk & 1][want];
        const uint64_t* rowp = mask + (size_t)(blk * 64 + r) * col_blocks;
#pragma unroll
        for (int k = 0; k < WPL; ++k) {
          const int w = st.base + k * 64 + lane;
          if (w < col_blocks) st.v[j][k] = rowp[w];
        }
      }
    }
  };
  for (int it = 0; it < col_blocks; ++it) {
    if (wv == 0) {
      const uint64_t word = diag;
      uint64_t cur[NMS_CARRY];
#pragma unroll
      for (int l = 0; l < NMS_CARRY; ++l) cur[l] = nx[l];
      const int cur_ord = ord;
      if (it + 1 < col_blocks) wave0_fetch(it + 1);
      const int rows_here = min(N - it * 64, 64);
      const uint64_t valid = rows_here >= 64 ? ~0ULL : ((1ULL << rows_here) - 1ULL);
      const uint64_t in = word & ((1ULL << lane) - 1ULL);   // earlier boxes of the tile that suppress this one
      uint64_t R = remv[it] | ~valid, K = 0;
      for (;;) {
        const uint64_t und = ~(R | K);
        if (!und) break;
        const bool mine = (und >> lane) & 1ULL;
        const bool rm = mine && (in & K) != 0;
        const bool kp = mine && !rm && (in & ~R) == 0;
        K |= __ballot(kp);
        R |= __ballot(rm);
      }
      if ((K >> lane) & 1ULL) {
#pragma unroll
        for (int l = 0; l < NMS_CARRY; ++l)
          if (cur[l]) atomicOr(&remv[it + 1 + l], (unsigned long long)cur[l]);
      }
      if (lane < rows_here) keep[cur_ord] = (int)((K >> lane) & 1ULL);
      if ((K >> lane) & 1ULL) keptrow[it & 1][__popcll(K & ((1ULL << lane) - 1ULL))] = lane;
      if (lane == 0) keptcnt[it & 1] = __popcll(K);
    } else {
      if (it & 1) helper_step(setB, it - 1); else helper_step(setA, it - 1);
    }
    // LDS traffic of this iteration done, then a raw barrier: __syncthreads() would also wait for the global loads
    // that are meant to stay in flight across it
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
}

// Grouped NMS (one call for all classes: box_results_with_nms_and_limit, core/test.py:862-879): groups are contiguous
// runs of the sorted order and never interact, so they are reduced in parallel.  Workgroup b (one wave) owns the groups
// that START inside block b - the positions from its first start up to the first start in a later block - and walks
// only their blocks with the resolve of nms_reduce_kernel; kept boxes OR their words for the later blocks that can still
// hold their group into a bitmap in LDS.  35 classes x <= 300 boxes: chains of <= 5-6 blocks instead of one of 80.
__global__ __launch_bounds__(64) void nms_reduce_grouped_kernel(const uint64_t* __restrict__ mask,
                                                                const int32_t* __restrict__ order,
                                                                const int32_t* __restrict__ sgroup, int N, int col_blocks,
                                                                int32_t* __restrict__ keep) {
  __shared__ unsigned long long remv[NMS_MAX_BOXES / 64];
  const int lane = threadIdx.x, b = blockIdx.x;
  auto starts = [&](int blk) {   // lanes whose box opens a group
    const int p = blk * 64 + lane;
    const bool st = p < N && (p == 0 || sgroup[p - 1] != sgroup[p]);
    return (uint64_t)__ballot(st);
  };
  const uint64_t s0 = starts(b);
  if (!s0) return;
  for (int i = lane; i < NMS_MAX_BOXES / 64; i += 64) remv[i] = 0ULL;
  __syncthreads();
  uint64_t own = ~0ULL << __builtin_ctzll(s0);   // lanes of the current block that belong to this workgroup's groups
  for (int it = b; it < col_blocks; ++it) {
    bool last = false;
    if (it > b) {
      const uint64_t st = starts(it);
      own = ~0ULL;
      if (st) {
        const int e = __builtin_ctzll(st);
        if (e == 0) break;
        own = (1ULL << e) - 1ULL;
        last = true;
      }
    }
    const int row = it * 64 + lane;
    const int rows_here = min(N - it * 64, 64);
    const uint64_t valid = rows_here >= 64 ? ~0ULL : ((1ULL << rows_here) - 1ULL);
    own &= valid;
    const uint64_t word = mask[(size_t)row * col_blocks + it];
    const uint64_t in = word & ((1ULL << lane) - 1ULL);
    uint64_t R = remv[it] | ~own, K = 0;
    for (;;) {
      const uint64_t und = ~(R | K);
      if (!und) break;
      const bool mine = (und >> lane) & 1ULL;
      const bool rm = mine && (in & K) != 0;
      const bool kp = mine && !rm && (in & ~R) == 0;
      K |= __ballot(kp);
      R |= __ballot(rm);
    }
    if ((own >> lane) & 1ULL) keep[order[row]] = (int)((K >> lane) & 1ULL);
    if (last || rows_here < 64) break;
    if ((K >> lane) & 1ULL) {
      const int g = sgroup[row];
      for (int c = it + 1; c < col_blocks && sgroup[c * 64] <= g; ++c) {
        const uint64_t w = mask[(size_t)row * col_blocks + c];
        if (w) atomicOr(&remv[c], (unsigned long long)w);
      }
    }
    __syncthreads();
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
  hipLaunchKernelGGL(nms_sort_kernel, dim3(col_blocks), dim3(1024), 0, s, boxes, scores, N, npad, ws.order, ws.sboxes,
                     ws.sareas, group, ws.sgroup);
  hipLaunchKernelGGL(nms_mask_kernel, dim3(col_blocks, col_blocks), dim3(64), 0, s, ws.sboxes,
                     ws.sareas, N, col_blocks, thresh, mode, ws.sgroup, group ? 1 : 0, ws.mask);
  if (group)
    hipLaunchKernelGGL(nms_reduce_grouped_kernel, dim3(col_blocks), dim3(64), 0, s, ws.mask, ws.order, ws.sgroup, N,
                       col_blocks, ws.keep);
  else if (col_blocks <= 128 + 1 + NMS_CARRY)
    hipLaunchKernelGGL(nms_reduce_kernel<2>, dim3(1), dim3(1024), 0, s, ws.mask, ws.order, N, col_blocks, ws.keep);
  else
    hipLaunchKernelGGL(nms_reduce_kernel<4>, dim3(1), dim3(1024), 0, s, ws.mask, ws.order, N, col_blocks, ws.keep);
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
