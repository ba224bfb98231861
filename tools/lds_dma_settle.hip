// Probe (diagnostic, not product code): after the ISSUING wave's `s_waitcnt vmcnt(0)` and an `s_barrier`, how soon may
// ANOTHER wave's ds_read of LDS-DMA (global_load_lds_dwordx4) data still return the OLD LDS bytes?
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/lds_dma_settle tools/lds_dma_settle.hip && tools/lds_dma_settle
//
// Every workgroup (4 waves) loops: each wave issues NP 1-KiB pieces into ring stage it & 1, waits vmcnt(0), passes a raw
// s_barrier, idles DELAY x 16 cycles, then reads the LAST piece of the NEXT wave and compares it with the same bytes
// loaded straight from global memory.  A mismatch is classified: "stale" = the bytes the stage held two iterations ago.
// Configurations: source footprint (L2-resident / HBM), workgroups per CU, a concurrent LDS-hammering kernel or a second
// instance of the probe on another stream.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x)                                                                      \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16(const void* g, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)lds_wave_base, 16, 0, 0);
}

__global__ void fill_kernel(uint4* p, long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long st = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += st) {
    const uint32_t j = (uint32_t)i;
    p[i] = make_uint4(j, j * 2654435761u, ~j, j ^ 0x5bd1e995u);
  }
}

template <int NP, int DELAY>
__global__ __launch_bounds__(256) void probe_kernel(const uint4* src, long long mask16, int iters, unsigned* counters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int STAGE = 4 * NP * 1024;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int w2 = (wave + 1) & 3;
  unsigned bad = 0, stale = 0;
  auto index = [&](int it, int w, int p) -> long long {
    return ((((long long)blockIdx.x * iters + it) * 4 + w) * NP + p) * 64 * 97 & mask16;   // scattered 1-KiB pieces
  };
  for (int it = 0; it < iters; ++it) {
    const int stage = it & 1;
    const uint4 want = src[index(it, w2, NP - 1) + lane];
    const uint4 old = it >= 2 ? src[index(it - 2, w2, NP - 1) + lane] : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int p = 0; p < NP; ++p) glds16(src + index(it, wave, p) + lane, smem + stage * STAGE + (wave * NP + p) * 1024);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int d = 0; d < DELAY; ++d) asm volatile("s_nop 15" ::: "memory");
    const uint4 got = *reinterpret_cast<const uint4*>(smem + stage * STAGE + (w2 * NP + NP - 1) * 1024 + lane * 16);
    if (got.x != want.x || got.y != want.y || got.z != want.z || got.w != want.w) {
      ++bad;
      if (got.x == old.x && got.y == old.y && got.z == old.z && got.w == old.w) ++stale;
    }
  }
  if (bad) {
    atomicAdd(&counters[0], bad);
    atomicAdd(&counters[1], stale);
    atomicAdd(&counters[2], 1u);
  }
}

// LDS traffic only: keeps the CU's LDS pipeline busy beside the probe
__global__ __launch_bounds__(256) void lds_hammer_kernel(int iters, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* s = reinterpret_cast<uint4*>(smem);
  const int tid = threadIdx.x;
  for (int i = tid; i < 1024; i += 256) s[i] = make_uint4(i, i, i, i);
  __syncthreads();
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const uint4 v = s[(tid + 37 * u + it) & 1023];
      acc.x += v.x; acc.y ^= v.y; acc.z += v.z; acc.w ^= v.w;
    }
    s[(tid * 5 + it) & 1023] = acc;
  }
  if (acc.x == 0x12345678u) sink[0] = acc.y;
}

struct Result { unsigned bad, stale, wgs; };

template <int NP, int DELAY>
static Result run(const uint4* src, long long n16, int wgs, int iters, int partner, unsigned* counters, hipStream_t s0, hipStream_t s1) {
  CHECK(hipMemsetAsync(counters, 0, 64, s0));
  CHECK(hipStreamSynchronize(s0));
  const int lds = 2 * 4 * NP * 1024;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&probe_kernel<NP, DELAY>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  long long mask = 1;
  while (mask * 2 <= n16 - 64 * 97) mask *= 2;
  mask = (mask - 1) & ~63LL;
  if (partner == 1) hipLaunchKernelGGL(lds_hammer_kernel, dim3(1024), dim3(256), 16384, s1, 20000, counters + 8);
  if (partner == 2) hipLaunchKernelGGL((probe_kernel<NP, DELAY>), dim3(wgs), dim3(256), lds, s1, src, mask, iters, counters + 4);
  hipLaunchKernelGGL((probe_kernel<NP, DELAY>), dim3(wgs), dim3(256), lds, s0, src, mask, iters, counters);
  CHECK(hipGetLastError());
  CHECK(hipStreamSynchronize(s0));
  CHECK(hipStreamSynchronize(s1));
  unsigned h[8];
  CHECK(hipMemcpy(h, counters, sizeof(h), hipMemcpyDeviceToHost));
  return Result{h[0] + h[4], h[1] + h[5], h[2] + h[6]};
}

template <int NP>
static void sweep(const uint4* src, long long n16_small, long long n16_big, unsigned* counters, hipStream_t s0, hipStream_t s1) {
  const char* pn[3] = {"alone", "beside an LDS hammer", "beside a second probe"};
  for (int big = 0; big < 2; ++big)
    for (int partner = 0; partner < 3; ++partner) {
      const long long n16 = big ? n16_big : n16_small;
      const int wgs = 2048, iters = 400;
      Result r[5];
      r[0] = run<NP, 0>(src, n16, wgs, iters, partner, counters, s0, s1);
      r[1] = run<NP, 2>(src, n16, wgs, iters, partner, counters, s0, s1);
      r[2] = run<NP, 8>(src, n16, wgs, iters, partner, counters, s0, s1);
      r[3] = run<NP, 32>(src, n16, wgs, iters, partner, counters, s0, s1);
      r[4] = run<NP, 128>(src, n16, wgs, iters, partner, counters, s0, s1);
      const double reads = (double)wgs * iters * 4 * (partner == 2 ? 2 : 1);
      printf("pieces/wave %2d  source %-8s  %-22s  wave-reads %.0f  mismatching lane-reads (stale) at delay 0/32/128/512/2048 cycles:",
             NP, big ? "HBM 1GiB" : "L2 8MiB", pn[partner], reads);
      for (int i = 0; i < 5; ++i) printf("  %u (%u)", r[i].bad, r[i].stale);
      printf("\n");
      fflush(stdout);
    }
}

int main() {
  const long long n16_big = (1LL << 30) / 16, n16_small = (8LL << 20) / 16;
  uint4* src;
  unsigned* counters;
  CHECK(hipMalloc(&src, n16_big * 16));
  CHECK(hipMalloc(&counters, 256));
  hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, src, n16_big);
  CHECK(hipDeviceSynchronize());
  hipStream_t s0, s1;
  CHECK(hipStreamCreate(&s0));
  CHECK(hipStreamCreate(&s1));
  sweep<1>(src, n16_small, n16_big, counters, s0, s1);
  sweep<4>(src, n16_small, n16_big, counters, s0, s1);
  sweep<8>(src, n16_small, n16_big, counters, s0, s1);
  printf("done\n");
  return 0;
}
