// Probe (diagnostic, not product code): what read rate does HBM give a weight-streaming kernel on this GPU, and does
// the ORDER in which the workgroups walk the matrix matter?
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/stream_probe tools/stream_probe.hip && tools/stream_probe
//
// 616 MB (fc1's weight matrix: 24 cout tiles x 1568 K-steps x 16 KiB blocks) is read once by 504 workgroups (24 tiles x
// 21 K-slices of 75 K-steps, as the FC launches them, two per CU on 252 CUs) with 16-byte loads, DEPTH blocks in flight
// per workgroup.  Layouts of the blocks in memory:
//   tile-major   block (tile, ks) at tile * 1568 + ks        (what ops.tile_weights produces)
//   step-major   block (tile, ks) at ks * 24 + tile          (the 24 tiles of a K-step adjacent)
//   linear       workgroup g reads blocks g, g + 504, ...    (upper bound: perfectly interleaved streams)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x)                                                                      \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

constexpr int TILES = 24, KSTEPS = 1568, SLICES = 21, PER = 75;   // 21 * 75 = 1575 >= 1568
constexpr int BLOCK = 16384;

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// MODE 3: tile-major order with NON-TEMPORAL loads (do streaming hints change the rate behind dirty caches?)
template <int MODE, int DEPTH>
__global__ __launch_bounds__(256) void stream_kernel(const uint4* __restrict__ w, uint4* __restrict__ sink) {
  const int g = blockIdx.x;
  const int tile = g % TILES, slice = g / TILES;
  const int ks0 = slice * PER, ks1 = min(ks0 + PER, KSTEPS);
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (int ks = ks0; ks < ks1; ks += DEPTH) {
    uint4 v[DEPTH][4];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int k = min(ks + d, ks1 - 1);
      long long blk;
      if (MODE == 0 || MODE == 3) blk = (long long)tile * KSTEPS + k;
      else if (MODE == 1) blk = (long long)k * TILES + tile;
      else blk = (long long)(k - ks0) * (TILES * SLICES) + g;
      if (MODE == 2 && blk >= (long long)TILES * KSTEPS) blk = g;
      const uint4* p = w + blk * (BLOCK / 16) + threadIdx.x;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (MODE == 3) {
          const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + i * 256));
          v[d][i] = make_uint4(t.x, t.y, t.z, t.w);
        } else {
          v[d][i] = p[i * 256];
        }
      }
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc.x ^= v[d][i].x; acc.y ^= v[d][i].y; acc.z ^= v[d][i].z; acc.w ^= v[d][i].w;
      }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[g * 256 + threadIdx.x] = acc;
}

__global__ void fill_kernel(uint4* p, long long n, uint32_t seed);

// The FC's own fetch pattern without its arithmetic: per K-step a wave issues PIECES LDS-DMA loads of 1 KiB (4 = the
// weight block alone, 9 = weight block + a 160-row activation tile from an L2-resident buffer) into a 3-stage ring,
// waits for the step before last (counted vmcnt) and passes a barrier - the cadence of igemm3's loop.
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
// ACT 0: the activation tile comes from a small L2-resident buffer; 1: from a 160 x 100352 row-major matrix as the FC
// reads it (row m, K-step ks = 128 bytes at m * 200704 + ks * 128: 160 separate lines per K-step); 2: from the same
// 32 MB laid out K-step-major ([ks][160 rows][128 B]: one contiguous 20 KiB block per K-step)
template <int PIECES, int NT, int ACT = 0>
__global__ __launch_bounds__(256) void dma_ring_kernel(const uint4* __restrict__ w, const uint4* __restrict__ act,
                                                       uint4* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int STAGE = PIECES * 4 * 1024;
  const int g = blockIdx.x;
  int tile = g % TILES, slice = g / TILES, per = PER;
  if (ACT >= 3) {          // 32 K-slices of 49 K-steps (768 workgroups); ACT 4: slices pinned to XCDs (workgroup g runs on XCD g % 8)
    per = 49;
    if (ACT == 4) {
      const int q = g >> 3;
      slice = (g & 7) + 8 * (q / TILES);
      tile = q % TILES;
    }
  }
  const int ks0 = slice * per, ks1 = min(ks0 + per, KSTEPS);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  auto issue = [&](int ks, int stage) {
    const char* src = reinterpret_cast<const char*>(w + ((long long)tile * KSTEPS + ks) * (BLOCK / 16)) + wave * 1024 + lane * 16;
    char* dst = smem + stage * STAGE + wave * 1024;
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      const char* p;
      if (i < 4) {
        p = src + i * 4096;
      } else if (ACT == 0) {
        p = reinterpret_cast<const char*>(act) + ((ks * 5 + (i - 4)) % 1280) * 4096 + wave * 1024 + lane * 16;
      } else {
        const int row = (i - 4) * 32 + wave * 8 + (lane >> 3);   // 0..159
        p = ACT != 2 ? reinterpret_cast<const char*>(act) + (long long)row * (KSTEPS * 128) + ks * 128 + (lane & 7) * 16
                     : reinterpret_cast<const char*>(act) + ((long long)ks * 160 + row) * 128 + (lane & 7) * 16;
      }
      if (NT && i < 4) __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(dst + i * 4096), 16, 0, 2);
      else __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(dst + i * 4096), 16, 0, 0);
    }
  };
  issue(ks0, 0);
  if (ks0 + 1 < ks1) issue(ks0 + 1, 1);
  uint32_t acc = 0;
  int st = 0;
  for (int ks = ks0; ks < ks1; ++ks) {
    if (ks + 1 < ks1) {
      if (PIECES == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (ks + 2 < ks1) issue(ks + 2, (st + 2) % 3);
    acc ^= *reinterpret_cast<const uint32_t*>(smem + st * STAGE + threadIdx.x * 16);   // one word per lane: keeps the ring alive
    st = (st + 1) % 3;
  }
  if (acc == 0x12345678u) sink[g * 256 + threadIdx.x] = make_uint4(acc, 0, 0, 0);
}

template <int PIECES, int NT, int ACT = 0>
static void run_dma(const char* name, const uint4* w, const uint4* act, uint4* sink, uint4* flush, long long flush_n) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  CHECK(hipFuncSetAttribute((const void*)dma_ring_kernel<PIECES, NT, ACT>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * PIECES * 4096));
  float best = 1e30f;
  for (int r = 0; r < 6; ++r) {
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, flush, flush_n, (uint32_t)r);
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((dma_ring_kernel<PIECES, NT, ACT>), dim3(TILES * (ACT >= 3 ? 32 : SLICES)), dim3(256), 3 * PIECES * 4096, 0, w, act, sink);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (r) best = ms < best ? ms : best;
  }
  printf("%-34s best %7.1f us  %5.2f TB/s of weights\n", name, best * 1e3, (double)TILES * KSTEPS * BLOCK / (best * 1e-3) * 1e-12);
}

__global__ void fill_kernel(uint4* p, long long n, uint32_t seed) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long st = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += st) {
    const uint32_t j = (uint32_t)i * 2654435761u + seed;
    p[i] = make_uint4(j, j ^ 0x9e3779b9u, ~j, j * 31u);
  }
}

__global__ void read_kernel(const uint4* p, long long n, uint4* sink) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long st = (long long)gridDim.x * blockDim.x;
  uint4 a = make_uint4(0, 0, 0, 0);
  for (; i < n; i += st) {
    const uint4 v = p[i];
    a.x ^= v.x; a.y ^= v.y; a.z ^= v.z; a.w ^= v.w;
  }
  if ((a.x ^ a.y ^ a.z ^ a.w) == 0x12345678u) sink[threadIdx.x] = a;
}

static bool g_clean_flush = false;   // evict by READING the flush buffer: the caches then hold clean lines only

template <int MODE, int DEPTH>
static void run(const char* name, const uint4* w, uint4* sink, uint4* flush, long long flush_n) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  float best = 1e30f, sum = 0.f;
  const int reps = 6;
  for (int r = 0; r < reps; ++r) {
    if (g_clean_flush) hipLaunchKernelGGL(read_kernel, dim3(4096), dim3(256), 0, 0, flush, flush_n, sink);
    else hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, flush, flush_n, (uint32_t)r);   // evict L2 / MALL
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((stream_kernel<MODE, DEPTH>), dim3(TILES * SLICES), dim3(256), 0, 0, w, sink);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (r) { best = ms < best ? ms : best; sum += ms; }
  }
  const double bytes = (double)TILES * KSTEPS * BLOCK;
  printf("%-13s depth %d  best %7.1f us  mean %7.1f us  %5.2f TB/s\n", name, DEPTH, best * 1e3, sum / (reps - 1) * 1e3,
         bytes / (best * 1e-3) * 1e-12);
}

int main() {
  const long long n = (long long)TILES * KSTEPS * (BLOCK / 16);
  uint4 *w, *sink, *flush;
  const long long flush_n = (1LL << 30) / 16;
  CHECK(hipMalloc(&w, n * 16));
  CHECK(hipMalloc(&sink, TILES * SLICES * 256 * 16));
  CHECK(hipMalloc(&flush, flush_n * 16));
  hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, w, n, 7u);
  CHECK(hipDeviceSynchronize());
  hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, flush, flush_n, 3u);
  CHECK(hipDeviceSynchronize());
  for (int pass = 0; pass < 2; ++pass) {
  g_clean_flush = pass == 1;
  printf("-- caches evicted by %s 1 GiB before each timed launch\n", g_clean_flush ? "READING" : "WRITING");
  run<0, 2>("tile-major", w, sink, flush, flush_n);
  run<0, 4>("tile-major", w, sink, flush, flush_n);
  run<0, 8>("tile-major", w, sink, flush, flush_n);
  run<3, 2>("tile-major nt", w, sink, flush, flush_n);
  run<3, 4>("tile-major nt", w, sink, flush, flush_n);
  run<1, 2>("step-major", w, sink, flush, flush_n);
  run<1, 4>("step-major", w, sink, flush, flush_n);
  run<1, 8>("step-major", w, sink, flush, flush_n);
  run<2, 2>("linear", w, sink, flush, flush_n);
  run<2, 4>("linear", w, sink, flush, flush_n);
  run<2, 8>("linear", w, sink, flush, flush_n);
  }
  uint4* act;
  CHECK(hipMalloc(&act, 1280 * 4096));
  hipLaunchKernelGGL(fill_kernel, dim3(256), dim3(256), 0, 0, act, (long long)1280 * 256, 9u);
  CHECK(hipDeviceSynchronize());
  printf("-- LDS-DMA ring at the FC's cadence, no arithmetic (behind a writer)\n");
  run_dma<4, 0>("ring, weights only, plain", w, act, sink, flush, flush_n);
  run_dma<4, 1>("ring, weights only, nt", w, act, sink, flush, flush_n);
  run_dma<9, 0>("ring, weights + act tile, plain", w, act, sink, flush, flush_n);
  run_dma<9, 1>("ring, weights + act tile, nt", w, act, sink, flush, flush_n);
  uint4* act32;
  CHECK(hipMalloc(&act32, (size_t)160 * KSTEPS * 128));
  hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, act32, (long long)160 * KSTEPS * 8, 11u);
  CHECK(hipDeviceSynchronize());
  run_dma<9, 1, 1>("  act from 160 x 100352 row-major", w, act32, sink, flush, flush_n);
  run_dma<9, 1, 2>("  act K-step-major (20 KiB blocks)", w, act32, sink, flush, flush_n);
  run_dma<9, 1, 3>("  row-major act, 32 slices", w, act32, sink, flush, flush_n);
  run_dma<9, 1, 4>("  row-major act, 32 slices on XCDs", w, act32, sink, flush, flush_n);
  return 0;
}
