// Implicit-GEMM Conv3d / batched GEMM on bf16 MFMA for gfx950 (CDNA4).
//
// Reference behaviour: every mmcv ConvModule (Conv3d + eval BatchNorm3d + ReLU)
// of lib/i3d/resnet3d.py (:95-118, :250-283, :549-557, :721-730), the residual
// add + ReLU of Bottleneck3d.forward (:292-318), the 1x1x1 convs and the two
// matmuls of mmcv NonLocal3d (call site resnet3d.py:289), and every nn.Linear of
// the relation head (lib/modeling_rel/model_builder_rel.py:266-331,
// lib/combine_tree/cluster_tree*.py).
//
// Design (MI355X-first, not a translation of cuDNN calls):
//   * activations are channels-last bf16 (N,T,H,W,C): one output position's
//     receptive-field tap is a contiguous C-vector, so the im2col matrix is
//     never materialised - each K-step gathers 16-byte pieces straight from HBM/L2
//     into LDS by LDS-DMA (padding-halo pieces read a zero page);
//   * weights are pre-packed (Cout, taps*C) bf16 with the BN scale folded in, the
//     BN shift is the epilogue bias; residual add + ReLU are fused in the epilogue;
//   * D = W_tile (A operand, rows = Cout) x X_tile^T (B operand, cols = output
//     positions) on v_mfma_f32_16x16x32_bf16: a lane then owns 4 consecutive
//     output channels of one position (8 after the row permutation below) -> 16-byte stores;
//   * 4 waves / workgroup (2 x 2), wave tile (WM*16 couts) x (WN*16 positions),
//     BK = 64, NST-deep LDS ring, XOR-swizzled 16-byte chunks (conflict-free
//     ds_read_b128), the next K-step's LDS-DMA loads in flight during the MFMA block,
//     one barrier per K-step;
//   * the same kernel runs plain/batched GEMMs (1 tap, explicit row strides).
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace trace {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

struct IgemmParams {
  const uint16_t* in;
  const uint16_t* w;
  const float* bias;
  const uint16_t* res;
  void* out;
  int M;     // output positions (rows of the GEMM)
  int Cout;  // output channels (cols of the GEMM)
  int Ktot;  // kt*kh*kw*Crun
  int Crun;  // contiguous K elements per tap
  int To, Ho, Wo, Ti, Hi, Wi;
  int kt, kh, kw;
  int st, sh, sw, pt, ph, pw, dt, dh, dw;
  long long in_n_stride;  // elements between consecutive n (clips)
  int pix;                // elements between adjacent input W positions
  int ldw;                // weight row stride (elements)
  int ldo;                // output / residual row stride (elements)
  long long bs_in, bs_w, bs_out;  // blockIdx.z strides (elements)
  float alpha;
  int epi;
  int tiles_m, tiles_n;
  int m_fast;        // 1: position tile index fastest (weight-streaming shapes: few position tiles)
  int ksplit, ks_per;  // split-K: blockIdx.y = K slice of ks_per K-steps, raw f32 partials to `partial`
  float* partial;    // [ksplit][M][Cout] f32
  void* ws;          // caller workspace for split-K (may be null)
  int narrow;        // 1: Cout or ldo not a multiple of 8 -> 4-channel epilogue groups
  int dense;         // 1: rows of `in` are the GEMM rows (1 tap, unit stride, no padding)
  size_t ws_bytes;
  uint16_t* out2;    // optional second output: bf16(act2(out value)), same rows / ldo as `out`
  int epi2;          // TRACE_EPI_RELU | TRACE_EPI_LEAKY applied to the second output only
  long long bs_bias; // blockIdx.z stride of `bias` (elements)
  int xcd_slices;    // 1: split-K slices are pinned to XCDs (ksplit % 8 == 0), see igemm3_bf16_kernel
  const uint16_t* in2;  // dense GEMMs: second activation matrix - K-steps >= k_switch read row m of it (row stride pix2)
  int pix2, k_switch;
  int pool_t2;       // 1 (dense 128x128 tiles only): the epilogue max-pools frame pairs (MaxPool3d (2,1,1)/(2,1,1)), see igemm3
  const float* in_f32;  // stem16 only: the raw f32 clip (N,C,T,H,W) or (N,T,C,H,W) instead of the padded bf16 `in`
  int f_C, f_T, f_H, f_W, f_ntchw;
  const uint8_t* in_u8;  // stem16 only: uint8 frames (N,T,H,W,C), channels-last as decoded; value = (float)u8 - u8_mean[c]
  float u8_mean[3];
  int dry;           // 1: plan only (trace_*_workspace_bytes): launch nothing, leave the split-K scratch the plan wants in g_dry_bytes
  int w_tiled;       // 1: `w` is pre-tiled (TRACE_EPI_WTILED): [Cout/128][K/64] blocks of 128 rows x 64 k, LDS image order
  int w_stream;      // 1: every weight block is read by exactly one workgroup (one position tile): non-temporal loads
};

constexpr int BK = 64;

// byte offset of 16-byte chunk `chunk` (0..7) of row `r` in a [rows][64] bf16 tile
__device__ __forceinline__ int lds_off(int r, int chunk) {
  return r * 128 + ((chunk ^ (r & 7)) << 4);
}
// Weight (A-operand) tiles of the v3 / stem / tconv3 / sconv9 loops.  The MFMA rows of a weight fragment are the permuted
// couts perm_row(odd, fr) = {0-3, 8-11, 16-19, 24-27} (+ 4 for the odd fragment): under the (r & 7) key rows 0-3 and
// 24-27 (and 8-11 / 16-19) of one ds_read_b128 lane group ({0-3, 12-15, 20-27}, ... - MI355X_MICROARCH.md, LDS) fall on
// the same 16-byte slots of the 256-byte bank row, a 2-way conflict on every weight fragment read.  Row bits (0, 1, 3)
// as the key put the 16 lanes of every group on 16 different slots - for these rows and for consecutive rows alike.
__device__ __forceinline__ int a_key(int r) { return (r & 3) | ((r >> 1) & 4); }
__device__ __forceinline__ int lds_off_a(int r, int chunk) { return r * 128 + ((chunk ^ a_key(r)) << 4); }
// source-side chunk of lane `lane` of wave `wave` loading weight row (8 wave' + (lane >> 3)), wave' = wave (mod 2)
__device__ __forceinline__ int a_kseg(int wave, int lane) { return (lane & 7) ^ (((lane >> 3) & 3) | ((wave & 1) << 2)); }

// ---------------------------------------------------------------------------------------------
// v2 main kernel: both operands go HBM/L2 -> LDS with 16-byte LDS-DMA loads (no register staging, no
// ds_write pass); halo / tail pieces read a zero page instead of branching around the load; the
// XOR swizzle lives on the SOURCE address (LDS-DMA writes lane-linear) and on the fragment reads;
// weight rows are permuted inside a 32-cout group so that a lane ends up with 8 consecutive output
// channels of one position -> 16-byte channels-last stores / residual loads.
// ---------------------------------------------------------------------------------------------
// 4 KiB + one K-step: a halo piece's source pointer walks a whole tap (Crun <= 2048 elements) inside it
__device__ __attribute__((aligned(256))) const uint4 g_zero_page[264] = {};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16(const void* g, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)lds_wave_base, 16, 0, 0);
}
// The same with the NON-TEMPORAL cache policy (aux bit 1 = nt on gfx940+): for operands streamed from HBM exactly once -
// the weights of the 100352-wide FCs.  A plain load allocates in the memory-side cache and, behind kernels that
// wrote, pays for the write-back of the dirty lines it evicts: tools/stream_probe.hip reads fc1's 616 MB at 3.4 TB/s
// with plain loads behind a writer (6.1 TB/s on clean caches) and at 6.8 TB/s with nt loads in either state.
__device__ __forceinline__ void glds16_nt(const void* g, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)lds_wave_base, 16, 0, 2);
}

// cout offset (inside a 32-cout group) held by MFMA row `rho` of the even/odd fragment of a pair
__device__ __forceinline__ int perm_row(int odd, int rho) { return ((rho >> 2) << 3) + (odd << 2) + (rho & 3); }

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// The barrier of the LDS rings.  A wave must not ARRIVE with LDS reads still in flight: behind the barrier the other
// waves refill - by LDS-DMA - the very stage those reads came from (write-after-read).  `__builtin_amdgcn_s_barrier()`
// alone does not order them: hipcc keeps a stage's last fragment reads in flight across it and waits for them
// (`s_waitcnt lgkmcnt`) only in front of the MFMAs that it sank below the barrier.  Alone on a CU the reads always win the
// race against a refill that still has an L2 round trip ahead of it; beside another kernel's workgroups that keep the CU's
// LDS pipeline busy (a second stream or a forked graph branch) the refill can land first and the MFMA consumes bytes of
// the NEXT K-step: one wrong 32-position wave column per ~10^5 workgroups, found in round 3 (tools/race_probe.py named the
// launch, tools/scan_barrier_hazard.py lists every kernel whose assembly had the pattern).
__device__ __forceinline__ void ring_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Diagnostic build switch (tools/build_variant.py): idle for TRACE_SETTLE_SLEEP x 64 cycles between the barrier that follows
// a counted wait and the first fragment read of the stage that wait retired.
#ifdef TRACE_SETTLE_SLEEP
#define TRACE_SETTLE() asm volatile("s_sleep %0" ::"n"(TRACE_SETTLE_SLEEP) : "memory")
#else
#define TRACE_SETTLE() do {} while (0)
#endif

// Epilogue shared by the main-loop variants: bias / residual / activation / (second output) / split-K slab.
template <int WM, int WN>
__device__ __forceinline__ void igemm_epilogue_rows(const IgemmParams& p, f32x4 (&acc)[WM][WN], int cout0,
                                                    const int (&mrow)[WN], int wm, int fq, int kslice);

// positions of a tile are consecutive output rows: fragment j of wave column wn, lane fr -> row pos0 + ...
template <int WM, int WN>
__device__ __forceinline__ void igemm_epilogue(const IgemmParams& p, f32x4 (&acc)[WM][WN], int cout0, int pos0, int wm,
                                               int wn, int fr, int fq, int kslice) {
  int mrow[WN];
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    const int m = pos0 + wn * (16 * WN) + j * 16 + fr;
    mrow[j] = m < p.M ? m : -1;
  }
  igemm_epilogue_rows<WM, WN>(p, acc, cout0, mrow, wm, fq, kslice);
}

// mrow[j]: output row (position) of this lane's column of fragment j, or -1 when it lies outside the problem
template <int WM, int WN>
__device__ __forceinline__ void igemm_epilogue_rows(const IgemmParams& p, f32x4 (&acc)[WM][WN], int cout0,
                                                    const int (&mrow)[WN], int wm, int fq, int kslice) {
  // lane holds, for position fr of fragment j and pair q: couts cbase + q*32 + fq*8 + [0..7]
  // (acc[2q][j] = first four, acc[2q+1][j] = last four)
  const int cbase = cout0 + wm * (16 * WM) + fq * 8;
  if (p.ksplit > 1) {
    float* slab = p.partial + (long long)kslice * p.M * p.Cout;
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int m = mrow[j];
      if (m < 0) continue;
#pragma unroll
      for (int q = 0; q < WM / 2; ++q) {
        const int c = cbase + q * 32;
        if (c >= p.Cout) continue;
        float* o = slab + (long long)m * p.Cout + c;
        *reinterpret_cast<float4*>(o) = make_float4(acc[2 * q][j][0], acc[2 * q][j][1], acc[2 * q][j][2], acc[2 * q][j][3]);
        if (c + 4 < p.Cout)
          *reinterpret_cast<float4*>(o + 4) =
              make_float4(acc[2 * q + 1][j][0], acc[2 * q + 1][j][1], acc[2 * q + 1][j][2], acc[2 * q + 1][j][3]);
      }
    }
    return;
  }
  const bool relu = p.epi & TRACE_EPI_RELU;
  const bool leaky = p.epi & TRACE_EPI_LEAKY;
  const bool res_post = (p.epi & TRACE_EPI_RES_POST) && p.res;
  const bool has_res = (p.epi & TRACE_EPI_RESIDUAL) && p.res;
  const bool out_f32 = p.epi & TRACE_EPI_OUT_F32;
  const bool res_f32 = p.epi & TRACE_EPI_RES_F32;
  const float* bias = p.bias ? p.bias + (long long)blockIdx.z * p.bs_bias : nullptr;
  const long long obase = (long long)blockIdx.z * p.bs_out;
  // loads use clamped (always valid) addresses so that they need no per-lane branch; only the store
  // is predicated.  Cout % 8 == 0 is required by the host wrapper.
  long long orow[WN];
  bool mok[WN];
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    mok[j] = mrow[j] >= 0;
    orow[j] = obase + (long long)max(mrow[j], 0) * p.ldo;
  }
  if (p.narrow) {
    // N or ldc not a multiple of 8 (plain GEMMs only): 4-channel groups, 8-byte accesses
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      const int c = cbase + (i >> 1) * 32 + (i & 1) * 4;
      const bool cok = c < p.Cout;
      const int cl = min(c, p.Cout - 4);
      float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);
      if (bias) bq = *reinterpret_cast<const float4*>(bias + cl);
      const float bv[4] = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        float v[4], rf[4] = {0.f, 0.f, 0.f, 0.f};
        if (has_res || res_post) {
          if (res_f32) {
            const float4 r0 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.res) + orow[j] + cl);
            rf[0] = r0.x; rf[1] = r0.y; rf[2] = r0.z; rf[3] = r0.w;
          } else {
            const uint2 rv = *reinterpret_cast<const uint2*>(p.res + orow[j] + cl);
            rf[0] = __uint_as_float(rv.x << 16);
            rf[1] = __uint_as_float(rv.x & 0xffff0000u);
            rf[2] = __uint_as_float(rv.y << 16);
            rf[3] = __uint_as_float(rv.y & 0xffff0000u);
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = acc[i][j][r] * p.alpha + bv[r];
          if (has_res) v[r] += rf[r];
          if (relu) v[r] = fmaxf(v[r], 0.f);
          if (leaky) v[r] = v[r] > 0.f ? v[r] : 0.1f * v[r];
          if (res_post) v[r] += rf[r];
        }
        if (mok[j] && cok) {
          const long long o = orow[j] + c;
          if (out_f32) {
            *reinterpret_cast<float4*>((float*)p.out + o) = make_float4(v[0], v[1], v[2], v[3]);
          } else {
            uint2 ov;
            ov.x = pack_bf16x2(v[0], v[1]);
            ov.y = pack_bf16x2(v[2], v[3]);
            *reinterpret_cast<uint2*>((uint16_t*)p.out + o) = ov;
          }
          if (p.out2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              if (p.epi2 & TRACE_EPI_RELU) v[r] = fmaxf(v[r], 0.f);
              if (p.epi2 & TRACE_EPI_LEAKY) v[r] = v[r] > 0.f ? v[r] : 0.1f * v[r];
            }
            uint2 ov;
            ov.x = pack_bf16x2(v[0], v[1]);
            ov.y = pack_bf16x2(v[2], v[3]);
            *reinterpret_cast<uint2*>(p.out2 + o) = ov;
          }
        }
      }
    }
    return;
  }
#pragma unroll
  for (int q = 0; q < WM / 2; ++q) {
    const int c = cbase + q * 32;
    const bool cok = c < p.Cout;
    const int cl = min(c, p.Cout - 8);
    float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (bias) {
      const float4 b0 = *reinterpret_cast<const float4*>(bias + cl);
      const float4 b1 = *reinterpret_cast<const float4*>(bias + cl + 4);
      bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w;
      bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
    }
    float rf[WN][8];
    if (has_res || res_post) {
      if (res_f32) {
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          const float* rp = reinterpret_cast<const float*>(p.res) + orow[j] + cl;
          const float4 r0 = *reinterpret_cast<const float4*>(rp);
          const float4 r1 = *reinterpret_cast<const float4*>(rp + 4);
          rf[j][0] = r0.x; rf[j][1] = r0.y; rf[j][2] = r0.z; rf[j][3] = r0.w;
          rf[j][4] = r1.x; rf[j][5] = r1.y; rf[j][6] = r1.z; rf[j][7] = r1.w;
        }
      } else {
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          const uint4 rv = *reinterpret_cast<const uint4*>(p.res + orow[j] + cl);
          const uint32_t rw[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            rf[j][2 * r] = __uint_as_float(rw[r] << 16);
            rf[j][2 * r + 1] = __uint_as_float(rw[r] & 0xffff0000u);
          }
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int r = 0; r < 8; ++r) rf[j][r] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      float v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[2 * q][j][r] * p.alpha + bv[r];
        v[4 + r] = acc[2 * q + 1][j][r] * p.alpha + bv[4 + r];
      }
      if (has_res) {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] += rf[j][r];
      }
      if (relu) {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = fmaxf(v[r], 0.f);
      }
      if (leaky) {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = v[r] > 0.f ? v[r] : 0.1f * v[r];
      }
      if (res_post) {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] += rf[j][r];
      }
      if (mok[j] && cok) {
        const long long o = orow[j] + c;
        if (out_f32) {
          float* op = (float*)p.out + o;
          *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
          *reinterpret_cast<float4*>(op + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else {
          uint4 ov;
          ov.x = pack_bf16x2(v[0], v[1]);
          ov.y = pack_bf16x2(v[2], v[3]);
          ov.z = pack_bf16x2(v[4], v[5]);
          ov.w = pack_bf16x2(v[6], v[7]);
          *reinterpret_cast<uint4*>((uint16_t*)p.out + o) = ov;
        }
        if (p.out2) {
          if (p.epi2 & TRACE_EPI_RELU) {
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = fmaxf(v[r], 0.f);
          }
          if (p.epi2 & TRACE_EPI_LEAKY) {
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = v[r] > 0.f ? v[r] : 0.1f * v[r];
          }
          uint4 ov;
          ov.x = pack_bf16x2(v[0], v[1]);
          ov.y = pack_bf16x2(v[2], v[3]);
          ov.z = pack_bf16x2(v[4], v[5]);
          ov.w = pack_bf16x2(v[6], v[7]);
          *reinterpret_cast<uint4*>(p.out2 + o) = ov;
        }
      }
    }
  }
}

template <int WM, int WN, bool DENSE, int NST>
__global__ __launch_bounds__(256) void igemm2_bf16_kernel(const IgemmParams p) {
  static_assert(WM % 2 == 0, "cout fragments come in pairs");
  constexpr int TC = 32 * WM;  // cout rows per block
  constexpr int TP = 32 * WN;  // positions per block
  constexpr int A_BYTES = TC * 128;
  constexpr int B_BYTES = TP * 128;
  constexpr int STAGE = A_BYTES + B_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x;
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
  }
  const int tile_n = p.m_fast ? bid / p.tiles_m : bid % p.tiles_n;
  const int tile_m = p.m_fast ? bid % p.tiles_m : bid / p.tiles_n;
  const int cout0 = tile_n * TC;
  const int pos0 = tile_m * TP;

  const uint16_t* gin = p.in + (long long)blockIdx.z * p.bs_in;
  const uint16_t* gw = p.w + (long long)blockIdx.z * p.bs_w;
  const uint16_t* zero = reinterpret_cast<const uint16_t*>(g_zero_page);

  // ---- loader assignment: one LDS-DMA instruction of a wave fills 8 rows x 128 B; this lane owns
  // physical chunk (lane&7) of row (lane>>3) of it, i.e. LOGICAL k-chunk (lane&7)^(lane>>3)
  const int lrow = wave * 8 + (lane >> 3);      // row inside a 32-row round
  const int kseg = (lane & 7) ^ (lane >> 3);    // logical 16-byte chunk of the K-step

  const int ksteps_all = (p.Ktot + BK - 1) / BK;
  const int ks_begin = blockIdx.y * p.ks_per;
  const int ksteps = min(ksteps_all - ks_begin, p.ks_per);
  int klin = ks_begin * BK + kseg * 8;

  const uint16_t* wp[WM];
#pragma unroll
  for (int i = 0; i < WM; ++i) {
    // A-tile row (32*i + lrow) holds weight row cout0 + 32*(row>>5) + perm(row&31): the permutation is
    // applied when fragments are READ, the tile itself is in natural cout order
    int row = min(cout0 + lrow + 32 * i, p.Cout - 1);
    wp[i] = gw + (long long)row * p.ldw + klin;
  }
  // positions
  const uint16_t* xp[WN];   // DENSE: running pointers
  long long x_nbase[WN];
  int x_t0[WN], x_h0[WN], x_w0[WN];
#pragma unroll
  for (int i = 0; i < WN; ++i) {
    int m = pos0 + lrow + 32 * i;
    if (DENSE) {
      m = min(m, p.M - 1);
      xp[i] = gin + (long long)m * p.pix + klin;
    } else {
      if (m < p.M) {
        int wo = m % p.Wo;
        int t1 = m / p.Wo;
        int ho = t1 % p.Ho;
        int t2 = t1 / p.Ho;
        int to = t2 % p.To;
        int n = t2 / p.To;
        x_nbase[i] = (long long)n * p.in_n_stride;
        x_t0[i] = to * p.st - p.pt;
        x_h0[i] = ho * p.sh - p.ph;
        x_w0[i] = wo * p.sw - p.pw;
      } else {
        x_nbase[i] = 0;
        x_t0[i] = -0x40000000;
        x_h0[i] = 0;
        x_w0[i] = 0;
      }
    }
  }
  int kc = 0, ta = 0, tb = 0, tcw = 0;
  if (!DENSE) {
    const int tap = klin / p.Crun;
    kc = klin - tap * p.Crun;
    tcw = tap % p.kw;
    const int t2 = tap / p.kw;
    tb = t2 % p.kh;
    ta = t2 / p.kh;
  }

  auto issue = [&](int stage) {
    char* a = smem + stage * STAGE + wave * 1024;
    char* b = a + A_BYTES;
    const bool kvalid = klin < p.Ktot;
#pragma unroll
    for (int i = 0; i < WM; ++i) glds16(kvalid ? wp[i] : zero, a + i * 4096);
#pragma unroll
    for (int i = 0; i < WN; ++i) {
      const uint16_t* src;
      if (DENSE) {
        src = kvalid ? xp[i] : zero;
      } else {
        int ti = x_t0[i] + ta * p.dt, hi = x_h0[i] + tb * p.dh, wi = x_w0[i] + tcw * p.dw;
        bool ok = kvalid && (unsigned)ti < (unsigned)p.Ti && (unsigned)hi < (unsigned)p.Hi &&
                  (unsigned)wi < (unsigned)p.Wi;
        long long off = x_nbase[i] + ((long long)(ti * p.Hi + hi) * p.Wi + wi) * p.pix + kc;
        src = ok ? gin + off : zero;
      }
      glds16(src, b + i * 4096);
    }
  };
  auto advance = [&]() {
    klin += BK;
#pragma unroll
    for (int i = 0; i < WM; ++i) wp[i] += BK;
    if (DENSE) {
#pragma unroll
      for (int i = 0; i < WN; ++i) xp[i] += BK;
    } else {
      kc += BK;
      while (kc >= p.Crun) {
        kc -= p.Crun;
        if (++tcw == p.kw) {
          tcw = 0;
          if (++tb == p.kh) {
            tb = 0;
            ++ta;
          }
        }
      }
    }
  };

  f32x4 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fq = lane >> 4;
  // A fragment i (pair q = i>>1, odd = i&1), MFMA row fr  <-  tile row wm*16*WM + q*32 + perm_row(odd, fr)
  int a_row[WM];
#pragma unroll
  for (int i = 0; i < WM; ++i) a_row[i] = wm * (16 * WM) + (i >> 1) * 32 + perm_row(i & 1, fr);

  // NST-deep LDS ring: the loads of K-step ks+NST-1 are issued at step ks (into the stage read at step
  // ks-1), so NST-2 steps stay in flight across each barrier; the wait is a counted vmcnt (never 0 in
  // steady state) followed by a raw s_barrier (__syncthreads() would drain the LDS-DMA queue).
  constexpr int L = WM + WN;  // LDS-DMA instructions per thread per stage
#pragma unroll
  for (int s0 = 0; s0 < NST - 1; ++s0) {
    if (s0 < ksteps) {
      if (s0) advance();
      issue(s0);
    }
  }
  int cur = 0, nxt = NST - 1;
  for (int ks = 0; ks < ksteps; ++ks) {
    const int ahead = min(NST - 2, ksteps - 1 - ks);  // K-steps issued after ks that may stay in flight
    if (NST >= 4 && ahead >= 2) wait_vmcnt<2 * L>();
    else if (NST >= 3 && ahead == 1) wait_vmcnt<L>();
    else wait_vmcnt<0>();
    ring_barrier();  // step ks landed for every wave; every wave is done reading step ks-1's stage
    if (ks + NST - 1 < ksteps) {
      advance();
      issue(nxt);
    }
    TRACE_SETTLE();
    const char* a = smem + cur * STAGE;
    const char* b = a + A_BYTES;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 af[WM], bfr[WN];
#pragma unroll
      for (int i = 0; i < WM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(a + lds_off(a_row[i], s * 4 + fq));
#pragma unroll
      for (int j = 0; j < WN; ++j)
        bfr[j] = *reinterpret_cast<const bf16x8*>(b + lds_off(wn * (16 * WN) + j * 16 + fr, s * 4 + fq));
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    cur = (cur + 1 == NST) ? 0 : cur + 1;
    nxt = (nxt + 1 == NST) ? 0 : nxt + 1;
  }

  igemm_epilogue<WM, WN>(p, acc, cout0, pos0, wm, wn, fr, fq, blockIdx.y);
}


// ---------------------------------------------------------------------------------------------
// v3 main loop: same tiles, LDS image and epilogue as v2, but the K loop carries (almost) no address
// arithmetic.  Measured on v2 (rocprofv3 SQ counters, layer4 3x3 conv): a wave spent 611 cycles per K-step
// issuing instructions for 256 cycles of MFMA - per-piece bounds checks, 64-bit multiply-adds and LDS
// address computations - so the matrix pipe idled behind the scalar / vector ALU work.  Here
//   * a K-step never straddles a tap (Crun % 64 == 0): source pointers advance by 128 B per K-step and
//     are re-derived only when the tap changes (uniform branch), from a per-piece corner offset, a
//     per-piece tap-validity bit mask (both computed once) and a wave-uniform tap offset;
//   * halo / tail pieces point into the zero page and advance like every other piece (no select);
//   * fragment ds_read addresses are 6 per-lane constants + compile-time immediates: the K loop is
//     unrolled by the ring depth so that the stage base is an immediate too;
//   * a K tail (dense GEMMs with K % 64 != 0) is handled in the last K-step only.
// ---------------------------------------------------------------------------------------------
template <int WM, int WN, bool DENSE, int NST>
__global__ __launch_bounds__(256) void igemm3_bf16_kernel(const IgemmParams p) {
  static_assert(WM % 2 == 0, "cout fragments come in pairs");
  constexpr int TC = 32 * WM;
  constexpr int TP = 32 * WN;
  constexpr int A_BYTES = TC * 128;
  constexpr int B_BYTES = TP * 128;
  constexpr int STAGE = A_BYTES + B_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  int bid = blockIdx.x;
  int kslice = blockIdx.y;
  if (p.xcd_slices) {
    // split-K with ksplit % 8 == 0: workgroups go to XCDs round-robin in dispatch order, so give XCD x the slices
    // x, x+8, ... with ALL their tiles: the activation K-slice (not L2-resident as a whole: 32 MB for the 100352-wide
    // FCs) is then fetched into ONE L2 and shared by every cout tile, instead of once per XCD
    const int lin = blockIdx.y * gridDim.x + blockIdx.x;
    const int q = lin >> 3;
    kslice = (lin & 7) + 8 * (q / (int)gridDim.x);
    bid = q % (int)gridDim.x;
  } else {
    const int nwg = gridDim.x;
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
  }
  const int tile_n = p.m_fast ? bid / p.tiles_m : bid % p.tiles_n;
  const int tile_m = p.m_fast ? bid % p.tiles_m : bid / p.tiles_n;
  const int cout0 = tile_n * TC;
  const int pos0 = tile_m * TP;

  const uint16_t* gin = p.in + (long long)blockIdx.z * p.bs_in;
  const uint16_t* gw = p.w + (long long)blockIdx.z * p.bs_w;
  const char* zero = reinterpret_cast<const char*>(g_zero_page);

  const int lrow = wave * 8 + (lane >> 3);
  const int kseg = (lane & 7) ^ (lane >> 3);
  const int kseg_a = a_kseg(wave, lane);

  const int ksteps_all = (p.Ktot + BK - 1) / BK;
  const int ks_begin = kslice * p.ks_per;
  const int ksteps = min(ksteps_all - ks_begin, p.ks_per);
  const int kbase = ks_begin * BK;           // uniform
  const int klin0 = kbase + kseg * 8;        // this lane's first K element
  // K tail: only dense problems reach here with Ktot % 64 != 0; the tail K-step is the global last one
  const bool has_tail = DENSE && (p.Ktot & (BK - 1)) != 0 && (ks_begin + ksteps == ksteps_all);
  const bool lane_tail = (ksteps_all - 1) * BK + kseg * 8 >= p.Ktot;
  const bool lane_tail_a = (ksteps_all - 1) * BK + kseg_a * 8 >= p.Ktot;

  const char* ap[WM];
  int a_step = 2 * BK;     // bytes a weight pointer advances per K-step
  const bool w_stream = p.w_stream != 0;
  if (p.w_tiled) {
    // pre-tiled weights (TC == 128): K-step ks of cout tile tile_n is ONE contiguous 16 KiB block already in the
    // swizzled LDS image order, so a wave instruction reads 1 KiB of consecutive addresses (DRAM-page friendly:
    // the row-major layout touches 128 different weight rows, 200 KB apart, per K-step)
    a_step = A_BYTES;
#pragma unroll
    for (int i = 0; i < WM; ++i)
      ap[i] = reinterpret_cast<const char*>(gw) + ((long long)tile_n * ksteps_all + ks_begin) * A_BYTES +
              (i * 4 + wave) * 1024 + lane * 16;
  } else {
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      const int row = min(cout0 + lrow + 32 * i, p.Cout - 1);
      ap[i] = reinterpret_cast<const char*>(gw + (long long)row * p.ldw + kbase + kseg_a * 8);
    }
  }
  const char* bp[WN];
  const char* bp2[WN];     // dense, two sources: the rows of the second one
  int sw_left = (DENSE && p.in2) ? p.k_switch : 0x7fffffff;   // K-steps until the switch (uniform)
  long long boff[WN];      // conv: element offset of the receptive field's corner (+ this lane's k chunk)
  unsigned bmask[WN];      // conv: bit `tap` set <=> that tap of this position lies inside the input
  int tap = 0, tap_left = 0x7fffffff;
  int ta = 0, tb = 0, tcw = 0;
  // pool_t2 (WN == 4): a tile is 64 POOLED positions; wave column wn holds pooled positions 32 wn .. + 32 twice -
  // fragments 0, 1 from the even frame, fragments 2, 3 from the odd one - so the pair to be max-ed sits in ONE lane
  const bool pool2 = DENSE && WN == 4 && p.pool_t2;
  const int HW = p.Ho * p.Wo;
  if (DENSE) {
#pragma unroll
    for (int i = 0; i < WN; ++i) {
      int m = pos0 + lrow + 32 * i;
      if (pool2) {
        const int r = lrow + 32 * i;
        const int pp = min((pos0 >> 1) + (r >> 6) * 32 + ((r >> 4) & 1) * 16 + (r & 15), (p.M >> 1) - 1);
        m = pp + (pp / HW + ((r >> 5) & 1)) * HW;
      }
      m = min(m, p.M - 1);
      bp[i] = reinterpret_cast<const char*>(gin + (long long)m * p.pix + klin0);
      // second source (K = [K1 | K2] over two activation matrices, e.g. conv3(h) + downsample(x) of a bottleneck as
      // one GEMM): the same rows of `in2`, taken over from K-step k_switch on
      bp2[i] = p.in2 ? reinterpret_cast<const char*>(p.in2 + (long long)m * p.pix2 + kseg * 8) : nullptr;
    }
  } else {
    const int ntaps = p.kt * p.kh * p.kw;
#pragma unroll
    for (int i = 0; i < WN; ++i) {
      const int m = pos0 + lrow + 32 * i;
      unsigned mask = 0;
      long long off = 0;
      if (m < p.M) {
        const int wo = m % p.Wo;
        const int t1 = m / p.Wo;
        const int ho = t1 % p.Ho;
        const int t2 = t1 / p.Ho;
        const int to = t2 % p.To;
        const int n = t2 / p.To;
        const int t0 = to * p.st - p.pt, h0 = ho * p.sh - p.ph, w0 = wo * p.sw - p.pw;
        off = (long long)n * p.in_n_stride + ((long long)(t0 * p.Hi + h0) * p.Wi + w0) * p.pix + kseg * 8;
        int tp = 0;
        for (int a = 0; a < p.kt; ++a)
          for (int b = 0; b < p.kh; ++b)
            for (int c = 0; c < p.kw; ++c, ++tp) {
              const bool ok = (unsigned)(t0 + a * p.dt) < (unsigned)p.Ti && (unsigned)(h0 + b * p.dh) < (unsigned)p.Hi &&
                              (unsigned)(w0 + c * p.dw) < (unsigned)p.Wi;
              mask |= ok ? (1u << tp) : 0u;
            }
      }
      boff[i] = off;
      bmask[i] = mask;
    }
    (void)ntaps;
    const int steps_per_tap = p.Crun / BK;
    tap = __builtin_amdgcn_readfirstlane(ks_begin / steps_per_tap);
    const int in_tap = ks_begin - tap * steps_per_tap;
    tap_left = steps_per_tap - in_tap;
    tcw = tap % p.kw;
    const int t2 = tap / p.kw;
    tb = t2 % p.kh;
    ta = t2 / p.kh;
    const long long tap_off = ((long long)(ta * p.dt * p.Hi + tb * p.dh) * p.Wi + tcw * p.dw) * p.pix + in_tap * BK;
    const uint16_t* gtap = gin + tap_off;
#pragma unroll
    for (int i = 0; i < WN; ++i)
      bp[i] = ((bmask[i] >> tap) & 1u) ? reinterpret_cast<const char*>(gtap + boff[i]) : zero;
  }

  // wave-uniform tap walk: the tap's element offset moves by one of three precomputed increments
  const long long inc_w = (long long)p.dw * p.pix;
  const long long inc_h = ((long long)p.dh * p.Wi - (long long)p.kw * p.dw) * p.pix + inc_w;
  const long long inc_t = (((long long)p.dt * p.Hi - (long long)p.kh * p.dh) * p.Wi) * p.pix + inc_h;
  const uint16_t* gtap_cur = nullptr;
  if (!DENSE) gtap_cur = gin + ((long long)(ta * p.dt * p.Hi + tb * p.dh) * p.Wi + tcw * p.dw) * p.pix;
  const int steps_per_tap_u = DENSE ? 0 : p.Crun / BK;
  auto next_tap = [&]() {   // uniform: every lane of the workgroup changes tap at the same K-step
    ++tap;
    long long inc = inc_w;
    if (++tcw == p.kw) {
      tcw = 0;
      inc = inc_h;
      if (++tb == p.kh) {
        tb = 0;
        inc = inc_t;
      }
    }
    gtap_cur += inc;
    tap_left = steps_per_tap_u;
    const unsigned bit = 1u << tap;
#pragma unroll
    for (int i = 0; i < WN; ++i)
      bp[i] = (bmask[i] & bit) ? reinterpret_cast<const char*>(gtap_cur + boff[i]) : zero;
  };

  auto issue = [&](int stage, bool tail) {
    char* a = smem + stage * STAGE + wave * 1024;
    char* b = a + A_BYTES;
    if (tail) {   // dense only: lanes past the end of K read zeros
#pragma unroll
      for (int i = 0; i < WM; ++i) glds16(lane_tail_a ? zero : ap[i], a + i * 4096);
#pragma unroll
      for (int i = 0; i < WN; ++i) glds16(lane_tail ? zero : bp[i], b + i * 4096);
      return;
    }
    if (DENSE && w_stream) {   // uniform
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        glds16_nt(ap[i], a + i * 4096);
        ap[i] += a_step;
      }
    } else {
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        glds16(ap[i], a + i * 4096);
        ap[i] += a_step;
      }
    }
#pragma unroll
    for (int i = 0; i < WN; ++i) {
      glds16(bp[i], b + i * 4096);
      bp[i] += 2 * BK;
    }
    if (!DENSE) {
      if (--tap_left == 0) next_tap();
    } else if (--sw_left == 0) {
#pragma unroll
      for (int i = 0; i < WN; ++i) bp[i] = bp2[i];
    }
  };

  f32x4 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fq = lane >> 4;
  // per-lane fragment offsets inside a stage: [odd][s] for A (rows perm_row(odd, fr)), [s] for B (rows fr)
  int aoffs[2][2], boffs[2];
#pragma unroll
  for (int odd = 0; odd < 2; ++odd)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
      aoffs[odd][s2] = lds_off_a(wm * (16 * WM) + perm_row(odd, fr), s2 * 4 + fq);
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) boffs[s2] = A_BYTES + lds_off(wn * (16 * WN) + fr, s2 * 4 + fq);

  constexpr int L = WM + WN;
  int issued = 0;   // K-steps issued so far (uniform)
#pragma unroll
  for (int s0 = 0; s0 < NST - 1; ++s0) {
    if (s0 < ksteps) {
      issue(s0, has_tail && s0 == ksteps - 1);
      ++issued;
    }
  }
  int ks = 0;
  auto step = [&](int st) {   // st: ring stage of K-step ks (a constant in the unrolled main loop)
    const int ahead = issued - ks - 1;   // K-steps issued after ks
    if (NST >= 4 && ahead >= 2) wait_vmcnt<2 * L>();
    else if (NST >= 3 && ahead >= 1) wait_vmcnt<L>();
    else wait_vmcnt<0>();
    ring_barrier();
    if (issued < ksteps) {
      const int nst_stage = st + NST - 1;
      issue(nst_stage >= NST ? nst_stage - NST : nst_stage, has_tail && issued == ksteps - 1);
      ++issued;
    }
    TRACE_SETTLE();
    const char* sb = smem + st * STAGE;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      bf16x8 af[WM], bfr[WN];
#pragma unroll
      for (int i = 0; i < WM; ++i)
        af[i] = *reinterpret_cast<const bf16x8*>(sb + aoffs[i & 1][s2] + (i >> 1) * (32 * 128));
#pragma unroll
      for (int j = 0; j < WN; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(sb + boffs[s2] + j * (16 * 128));
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    ++ks;
  };
  // whole rounds of the ring with the stage as an immediate, then the remaining (< NST) steps
  for (; ks + NST <= ksteps;) {
#pragma unroll
    for (int st = 0; st < NST; ++st) step(st);
  }
  for (int st = 0; ks < ksteps; ++st) step(st);
  if (pool2) {
    if constexpr (WN == 4) {
      // out[pp] = max over the frame pair of act(conv + bias (+ residual)); bf16 rounding is monotonic, so this equals
      // the max-pool of the rounded conv output
      const bool relu = p.epi & TRACE_EPI_RELU;
      const bool has_res = (p.epi & TRACE_EPI_RESIDUAL) && p.res;
      const int Mp = p.M >> 1;
      uint16_t* outp = reinterpret_cast<uint16_t*>(p.out);
#pragma unroll
      for (int q = 0; q < WM / 2; ++q) {
        const int c = cout0 + wm * (16 * WM) + fq * 8 + q * 32;
        if (c >= p.Cout) continue;
        float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
          const float4 b0 = *reinterpret_cast<const float4*>(p.bias + c);
          const float4 b1 = *reinterpret_cast<const float4*>(p.bias + c + 4);
          bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w;
          bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
        }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int pp = (pos0 >> 1) + wn * 32 + jj * 16 + fr;
          if (pp >= Mp) continue;
          const long long m0 = pp + (long long)(pp / HW) * HW;
          float best[8];
#pragma unroll
          for (int par = 0; par < 2; ++par) {
            float v[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              v[r] = acc[2 * q][jj + 2 * par][r] * p.alpha + bv[r];
              v[4 + r] = acc[2 * q + 1][jj + 2 * par][r] * p.alpha + bv[4 + r];
            }
            if (has_res) {
              const uint4 rv = *reinterpret_cast<const uint4*>(p.res + (m0 + (long long)par * HW) * p.ldo + c);
              const uint32_t rw[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                v[2 * r] += __uint_as_float(rw[r] << 16);
                v[2 * r + 1] += __uint_as_float(rw[r] & 0xffff0000u);
              }
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
              if (relu) v[r] = fmaxf(v[r], 0.f);
              best[r] = par ? fmaxf(best[r], v[r]) : v[r];
            }
          }
          uint4 ov;
          ov.x = pack_bf16x2(best[0], best[1]);
          ov.y = pack_bf16x2(best[2], best[3]);
          ov.z = pack_bf16x2(best[4], best[5]);
          ov.w = pack_bf16x2(best[6], best[7]);
          *reinterpret_cast<uint4*>(outp + (long long)pp * p.ldo + c) = ov;
        }
      }
    }
    return;
  }
  igemm_epilogue<WM, WN>(p, acc, cout0, pos0, wm, wn, fr, fq, kslice);
}

// ---------------------------------------------------------------------------------------------
// Stem conv (3 -> 64, 5x7x7, spatial stride 2; lib/i3d/resnet3d.py:721-730) with the input patch resident in
// LDS.  As an implicit GEMM through the generic loop the stem re-fetches every input pixel ~12 times and the
// 143 KB weight matrix once per 128 positions (430 KB of L2 -> LDS traffic per 128 outputs: 205 us, operand
// delivery bound).  Here a workgroup owns a 16 x 16 tile of output positions of one output frame:
//   * its input patch (5 frames x 37 rows x 38 pixels x 4 channels bf16 = 56 KB) is loaded ONCE by LDS-DMA;
//   * a (kt, kh) tap is one MFMA K-block (8 pixels x 4 channels = 32): the B fragment of output row j is a
//     16-byte read at  lane_base + ((kt*37 + 2j + kh) * 304)  - one per-lane register plus immediates (the
//     18 K-steps are fully unrolled); consecutive lanes read consecutive 16-byte chunks (conflict-free);
//   * weights stream through a 3-stage LDS-DMA ring (8 KB per K-step) as in the generic loop;
//   * 2 x 2 waves: wave (wm, wn) owns couts 32 wm .. +32 and output rows 8 wn .. +8 (8 fragments of 16 columns).
// 199 KB of traffic per 256 outputs, 2 workgroups per CU.
// ---------------------------------------------------------------------------------------------
constexpr int ST_KT = 5, ST_KH = 7;
constexpr int ST_PR = 37;                 // patch rows  = 2*16 + 7 - 2
constexpr int ST_PC = 19;                 // 16-byte chunks per patch row = (2*16 + 8 - 2) pixels * 8 B / 16
constexpr int ST_ROWB = ST_PC * 16;       // 304
constexpr int ST_CHUNKS = ST_KT * ST_PR * ST_PC;   // 3515
constexpr int ST_PATCH = 14 * 256 * 16;    // 57344: the 14 load rounds of 256 lanes (the last round over-runs the patch with zeros)
constexpr int ST_ASTAGE = 64 * 128;
constexpr int ST_LDS = ST_PATCH + 3 * ST_ASTAGE;
constexpr int ST_KSTEPS = (ST_KT * ST_KH + 1) / 2;  // 18 K-steps of two taps

// F32IN: the patch is built from the caller's f32 clip (planar, unpadded) - converted to bf16, channel-padded to 4 and
// written with ds_write_b128 in the same LDS image as the LDS-DMA path - so the clip never makes the round trip through
// a padded channels-last bf16 copy (trace_clip_to_ndhwc_bf16: 154 MB read + 122 MB written per 8 clips, 73 us).
// INMODE 2: the same from uint8 frames (N,T,H,W,C) as the video decoder leaves them: (float)u8 - mean[c], rounded to bf16
// exactly as INMODE 1 rounds the f32 value (bit-identical outputs for the same pixels), a quarter of the bytes over
// PCIe and HBM (test_engine_rel.py:420-520 hands the network mean-subtracted f32 frames).
#ifdef TRACE_STEM_NT
#define STEM_LD(ptr) __builtin_nontemporal_load(ptr)
#else
#define STEM_LD(ptr) (*(ptr))
#endif
template <int INMODE>
__global__ __launch_bounds__(256) void stem16_bf16_kernel(const IgemmParams p) {
  constexpr bool F32IN = INMODE != 0;   // the patch is built by the lanes (not LDS-DMA)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fq = lane >> 4;
  const char* zero = reinterpret_cast<const char*>(g_zero_page);

  const int tiles_w = p.Wo >> 4, tiles_h = p.Ho >> 4;
  int bid = blockIdx.x;
  const int tw = bid % tiles_w;
  bid /= tiles_w;
  const int th = bid % tiles_h;
  bid /= tiles_h;
  const int to = bid % p.To, n = bid / p.To;
  const int r0 = th * 16, c0 = tw * 16;

  // ---- patch: chunk q of the flattened (frame, row, 16-byte chunk) list -> LDS byte q*16
  if (!F32IN) {
    const long long row_b = (long long)p.Wi * 8, frame_b = (long long)p.Hi * row_b;
    const char* src0 = reinterpret_cast<const char*>(p.in) +
                       (((long long)n * p.Ti + (long long)to * p.st) * p.Hi + 2 * r0) * row_b + (long long)c0 * 16;
#pragma unroll
    for (int rd = 0; rd < (ST_CHUNKS + 255) / 256; ++rd) {
      const int q = rd * 256 + tid;
      const int f = q / (ST_PR * ST_PC);
      const int rem = q - f * (ST_PR * ST_PC);
      const int row = rem / ST_PC;
      const int ch = rem - row * ST_PC;
      const char* src = q < ST_CHUNKS ? src0 + f * frame_b + row * row_b + ch * 16 : zero;
      glds16(src, smem + (rd * 256 + wave * 64) * 16);
    }
  }
  // ---- weights ring
  const int lrow = wave * 8 + (lane >> 3);
  const int kseg = a_kseg(wave, lane);
  const char* ap[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
    ap[i] = reinterpret_cast<const char*>(p.w + (long long)(lrow + 32 * i) * p.ldw + kseg * 8);
  const bool tail_lane = (ST_KSTEPS - 1) * BK + kseg * 8 >= p.Ktot;   // past the last tap: zeros
  auto issue_a = [&](int stage, bool last) {
    char* a = smem + ST_PATCH + stage * ST_ASTAGE + wave * 1024;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      glds16((last && tail_lane) ? zero : ap[i], a + i * 4096);
      ap[i] += 2 * BK;
    }
  };
  issue_a(0, false);
  issue_a(1, false);
  if (F32IN) {
    // halo (2 frames, 3 rows, 3 pixels) and the over-run of the last round are zeros; a chunk = 2 pixels x (3 + 1) channels
    const int T = p.f_T, H = p.f_H, W = p.f_W, C = p.f_C;
    const long long plane = (long long)H * W;
    const long long cstride = p.f_ntchw ? plane : (long long)T * plane;
    // loads go out unconditionally from clamped (always valid) addresses, seven chunks (42 loads) per batch, and are
    // zeroed afterwards where the chunk lies in the halo: a branch around a load makes the compiler wait for each one
    constexpr int NR = (ST_CHUNKS + 255) / 256;   // 14
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      float raw[NR / 2][2][3];
      bool okc[NR / 2][2];
#pragma unroll
      for (int r7 = 0; r7 < NR / 2; ++r7) {
        const int q = (half * (NR / 2) + r7) * 256 + tid;
        const int f = q / (ST_PR * ST_PC);
        const int rem = q - f * (ST_PR * ST_PC);
        const int row = rem / ST_PC;
        const int ch = rem - row * ST_PC;
        const int t = to * p.st + f - 2, y = 2 * r0 + row - 3, x0 = 2 * c0 + 2 * ch - 3;
        const bool ok = q < ST_CHUNKS && (unsigned)t < (unsigned)T && (unsigned)y < (unsigned)H;
        const int tc = min(max(t, 0), T - 1), yc = min(max(y, 0), H - 1);
        if (INMODE == 1) {
          const float* base = p.in_f32 + (p.f_ntchw ? ((long long)n * T + tc) * C * plane : ((long long)n * C * T + tc) * plane) +
                              (long long)yc * W;
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int x = x0 + e;
            okc[r7][e] = ok && (unsigned)x < (unsigned)W;
            const int xc = min(max(x, 0), W - 1);
#pragma unroll
            for (int c = 0; c < 3; ++c) raw[r7][e][c] = STEM_LD(base + (c < C ? c : 0) * cstride + xc);
          }
        } else {
          const uint8_t* base = p.in_u8 + ((((long long)n * T + tc) * H + yc) * W) * C;
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int x = x0 + e;
            okc[r7][e] = ok && (unsigned)x < (unsigned)W;
            const int xc = min(max(x, 0), W - 1);
#pragma unroll
            for (int c = 0; c < 3; ++c) raw[r7][e][c] = (float)base[(long long)xc * C + (c < C ? c : 0)] - p.u8_mean[c];
          }
        }
      }
#pragma unroll
      for (int r7 = 0; r7 < NR / 2; ++r7) {
        const int q = (half * (NR / 2) + r7) * 256 + tid;
        float v[2][3];
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int c = 0; c < 3; ++c) v[e][c] = (okc[r7][e] && c < C) ? raw[r7][e][c] : 0.f;
        uint4 o;
        o.x = pack_bf16x2(v[0][0], v[0][1]);
        o.y = pack_bf16x2(v[0][2], 0.f);
        o.z = pack_bf16x2(v[1][0], v[1][1]);
        o.w = pack_bf16x2(v[1][2], 0.f);
        *reinterpret_cast<uint4*>(smem + q * 16) = o;
      }
    }
    __syncthreads();
  }

  f32x4 acc[2][8];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int aoffs[2][2];
#pragma unroll
  for (int odd = 0; odd < 2; ++odd)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) aoffs[odd][s2] = ST_PATCH + lds_off_a(wm * 32 + perm_row(odd, fr), s2 * 4 + fq);
  const int lane_b = wn * (16 * ST_ROWB) + (fr + fq) * 16;
  bf16x8 rf[2][11];   // patch-row fragments of the current frame tap, by row parity (see the K loop)

#pragma unroll
  for (int ks = 0; ks < ST_KSTEPS; ++ks) {
    // Drain EVERY weight stage in flight, not only step ks's: a stage is then read one whole K-step after the wait
    // that retired it (step ks's stage was retired by the wait of step ks-1; the stage still loading during a step is
    // the one issued after its barrier).  With the counted wait this loop used first - vmcnt(2): "step ks's stage has
    // landed, the next may stay in flight", read right behind the barrier - waves read bytes of another wave's DMA
    // that had been counted done but were not in LDS yet: a few wrong output channels per 8 x 16 positions in a few of
    // 3136 workgroups, only at B = 8 / 32 frames, run-dependent (race screen: tests/test_i3d_gpu.py::
    // test_backbone_is_deterministic_at_bench_size).  The generic loops put six DMA issues between barrier and first
    // fragment read; this kernel only two.
#ifdef TRACE_STEM_COUNTED
    if (ks + 1 < ST_KSTEPS) wait_vmcnt<2>(); else wait_vmcnt<0>();
#else
    wait_vmcnt<0>();
#endif
    ring_barrier();
    if (ks + 2 < ST_KSTEPS) issue_a((ks + 2) % 3, ks + 2 == ST_KSTEPS - 1);
    TRACE_SETTLE();
    const int abase = (ks % 3) * ST_ASTAGE;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int tap = 2 * ks + s2;
      if (tap >= ST_KT * ST_KH) break;
      const int kt_i = tap / ST_KH, kh_i = tap % ST_KH;
      const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(smem + abase + aoffs[0][s2]);
      const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(smem + abase + aoffs[1][s2]);
      // output row j under tap kh reads patch row 2 j + kh = 2 (j + kh / 2) + (kh & 1): the eight fragments of tap kh
      // are those of tap kh - 2 moved up one row.  Each patch row of a frame is therefore read ONCE (21 reads per kt
      // instead of 56) and kept in registers while the taps slide over it (the loop is fully unrolled: rf[][] are
      // registers, dead entries are dropped); the K loop was LDS-read bound at 10 fragment reads per 16 MFMAs
      const int par = kh_i & 1, sl = kh_i >> 1;
#pragma unroll
      for (int u = (sl == 0 ? 0 : sl + 7); u < sl + 8; ++u)
        rf[par][u] = *reinterpret_cast<const bf16x8*>(smem + lane_b + (kt_i * ST_PR + 2 * u + par) * ST_ROWB);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, rf[par][j + sl], acc[0][j], 0, 0, 0);
        acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, rf[par][j + sl], acc[1][j], 0, 0, 0);
      }
    }
  }
  // ---- epilogue: lane holds couts 32 wm + 8 fq + [0..7] of position (row 8 wn + j, column fr)
  const int cbase = wm * 32 + fq * 8;
  float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (p.bias) {
    const float4 b0 = *reinterpret_cast<const float4*>(p.bias + cbase);
    const float4 b1 = *reinterpret_cast<const float4*>(p.bias + cbase + 4);
    bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w;
    bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
  }
  const bool relu = p.epi & TRACE_EPI_RELU;
  uint16_t* out = reinterpret_cast<uint16_t*>(p.out);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const long long m = (((long long)n * p.To + to) * p.Ho + r0 + 8 * wn + j) * p.Wo + c0 + fr;
    float v[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v[r] = acc[0][j][r] + bv[r];
      v[4 + r] = acc[1][j][r] + bv[4 + r];
    }
    if (relu) {
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = fmaxf(v[r], 0.f);
    }
    uint4 ov;
    ov.x = pack_bf16x2(v[0], v[1]);
    ov.y = pack_bf16x2(v[2], v[3]);
    ov.z = pack_bf16x2(v[4], v[5]);
    ov.w = pack_bf16x2(v[6], v[7]);
    *reinterpret_cast<uint4*>(out + m * 64 + cbase) = ov;
  }
}

// Sum the split-K slabs in slice order (deterministic) and apply the epilogue.
__global__ __launch_bounds__(256) void igemm_splitk_reduce_kernel(const IgemmParams p) {
  const int cq = p.Cout >> 2;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)p.M * cq) return;
  const int m = (int)(idx / cq), c = (int)(idx % cq) * 4;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  // eight slabs' loads in flight per lane; the sum keeps the slab order (bit-identical to the one-by-one loop)
  const float* src = p.partial + (long long)m * p.Cout + c;
  const long long zs = (long long)p.M * p.Cout;
  int z = 0;
  for (; z + 8 <= p.ksplit; z += 8) {
    float4 t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const float4*>(src + (z + u) * zs);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      v[0] += t[u].x;
      v[1] += t[u].y;
      v[2] += t[u].z;
      v[3] += t[u].w;
    }
  }
  for (; z < p.ksplit; ++z) {
    const float4 t = *reinterpret_cast<const float4*>(src + z * zs);
    v[0] += t.x;
    v[1] += t.y;
    v[2] += t.z;
    v[3] += t.w;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] *= p.alpha;
  if (p.bias) {
    const float4 bv = *reinterpret_cast<const float4*>(p.bias + c);
    v[0] += bv.x;
    v[1] += bv.y;
    v[2] += bv.z;
    v[3] += bv.w;
  }
  const long long o = (long long)m * p.ldo + c;
  auto add_res = [&]() {
    if (p.epi & TRACE_EPI_RES_F32) {
      const float4 r0 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.res) + o);
      v[0] += r0.x;
      v[1] += r0.y;
      v[2] += r0.z;
      v[3] += r0.w;
    } else {
      const uint2 rv = *reinterpret_cast<const uint2*>(p.res + o);
      v[0] += __uint_as_float(rv.x << 16);
      v[1] += __uint_as_float(rv.x & 0xffff0000u);
      v[2] += __uint_as_float(rv.y << 16);
      v[3] += __uint_as_float(rv.y & 0xffff0000u);
    }
  };
  if ((p.epi & TRACE_EPI_RESIDUAL) && p.res) add_res();
  if (p.epi & TRACE_EPI_RELU) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
  }
  if (p.epi & TRACE_EPI_LEAKY) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.1f * v[r];
  }
  if ((p.epi & TRACE_EPI_RES_POST) && p.res) add_res();
  if (p.epi & TRACE_EPI_OUT_F32) {
    *reinterpret_cast<float4*>((float*)p.out + o) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
    uint2 ov;
    ov.x = pack_bf16x2(v[0], v[1]);
    ov.y = pack_bf16x2(v[2], v[3]);
    *reinterpret_cast<uint2*>((uint16_t*)p.out + o) = ov;
  }
  if (p.out2) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (p.epi2 & TRACE_EPI_RELU) v[r] = fmaxf(v[r], 0.f);
      if (p.epi2 & TRACE_EPI_LEAKY) v[r] = v[r] > 0.f ? v[r] : 0.1f * v[r];
    }
    uint2 ov;
    ov.x = pack_bf16x2(v[0], v[1]);
    ov.y = pack_bf16x2(v[2], v[3]);
    *reinterpret_cast<uint2*>(p.out2 + o) = ov;
  }
}

// ---------------------------------------------------------------------------------------------
// Weight-streaming GEMM (the 100352-wide FC layers of the box / subject / object heads and fc_r,
// fast_rcnn_heads.py:94-152, cluster_tree.py:369-372: a few hundred rois against 0.2 - 0.6 GB of weights).
// HBM-bound on the weights, so the design is about bytes in flight, not MFMA rate:
//   * 4 waves split the 128 couts of a workgroup (4 x 1 layout): a weight row is used by ONE wave, so
//     weights bypass LDS - each lane loads its MFMA A-fragments straight from HBM with 16-byte loads,
//     two K-steps ahead (3 register sets), 16 KB in flight per wave;
//   * the activations (all M <= 192 positions, L2-resident) go through a 3-stage LDS-DMA ring shared by
//     the four waves;
//   * K is split over blockIdx.y; slices write f32 slabs, igemm_splitk_reduce_kernel sums them in order.
// vmcnt bookkeeping: step s issues ONE group = {A fragments of step s+2, activation pieces of step s+2};
// the wait before step s leaves the newest group in flight.
// ---------------------------------------------------------------------------------------------
template <int WN>
__global__ __launch_bounds__(256) void wsgemm_bf16_kernel(const IgemmParams p) {
  constexpr int TP = 16 * WN;           // positions held by the workgroup (>= M)
  constexpr int NP = TP / 32;           // LDS-DMA pieces per thread per K-step
  constexpr int STAGE = TP * 128;
  constexpr int NSTB = 3;
  constexpr int GROUP = 4 + NP;         // vector-memory instructions per issue group
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int cout0 = blockIdx.x * 128 + wave * 32;
  const int ksteps_all = p.Ktot / BK;
  const int ks_begin = blockIdx.y * p.ks_per;
  const int ksteps = min(ksteps_all - ks_begin, p.ks_per);
  const long long kbase = (long long)ks_begin * BK;

  // A fragments: lane (fr, fq) of fragment i holds weights[cout0 + 16 i + fr][k + 32 s + 8 fq .. +8]
  const char* ap[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = min(cout0 + 16 * i + fr, p.Cout - 1);
    ap[i] = reinterpret_cast<const char*>(p.w + (long long)row * p.ldw + kbase + fq * 8);
  }
  // activations: same loader as the igemm tiles (8 rows x 128 B per wave instruction, XOR-swizzled chunks)
  const int lrow = wave * 8 + (lane >> 3);
  const int kseg = (lane & 7) ^ (lane >> 3);
  const char* bp[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int m = min(lrow + 32 * i, p.M - 1);
    bp[i] = reinterpret_cast<const char*>(p.in + (long long)m * p.pix + kbase + kseg * 8);
  }
  bf16x8 areg[3][2][2];   // [register set][fragment][K half]
  auto issue = [&](int set, int stage) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      areg[set][i][0] = *reinterpret_cast<const bf16x8*>(ap[i]);
      areg[set][i][1] = *reinterpret_cast<const bf16x8*>(ap[i] + 64);
      ap[i] += 2 * BK;
    }
    char* b = smem + stage * STAGE + wave * 1024;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      glds16(bp[i], b + i * 4096);
      bp[i] += 2 * BK;
    }
  };
  f32x4 acc[2][WN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int boffs[2];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) boffs[s2] = lds_off(fr, s2 * 4 + fq);

  int issued = 0, ks = 0;
  if (0 < ksteps) { issue(0, 0); ++issued; }
  if (1 < ksteps) { issue(1, 1); ++issued; }
  auto step = [&](int st) {   // st = ks % 3: register set and LDS stage of this K-step (compile-time in the callers)
    if (issued - ks - 1 >= 1) wait_vmcnt<GROUP>();
    else wait_vmcnt<0>();
    ring_barrier();
    if (issued < ksteps) {
      issue((st + 2) % 3, (st + 2) % 3);
      ++issued;
    }
    TRACE_SETTLE();
    const char* sb = smem + st * STAGE;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(sb + boffs[s2] + j * (16 * 128));
        acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(areg[st][0][s2], bfr, acc[0][j], 0, 0, 0);
        acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(areg[st][1][s2], bfr, acc[1][j], 0, 0, 0);
      }
    }
    ++ks;
  };
  for (; ks + 3 <= ksteps;) {
    step(0);
    step(1);
    step(2);
  }
  if (ks < ksteps) step(0);
  if (ks < ksteps) step(1);

  // f32 slab of this K slice: lane holds couts cout0 + 16 i + 4 fq + [0..3] of position 16 j + fr
  float* slab = p.partial + (long long)blockIdx.y * p.M * p.Cout;
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    const int m = j * 16 + fr;
    if (m >= p.M) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = cout0 + 16 * i + 4 * fq;
      if (c < p.Cout)
        *reinterpret_cast<float4*>(slab + (long long)m * p.Cout + c) =
            make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    }
  }
}

static thread_local size_t g_dry_bytes = 0;

template <int WN>
static int launch_wsgemm(IgemmParams& p, hipStream_t s, int ksplit_req) {
  const int ksteps = p.Ktot / BK;
  const int tiles_n = ceil_div(p.Cout, 128);
  long long ks = ksplit_req > 0 ? ksplit_req : (512 + tiles_n / 2) / tiles_n;
  if (ks > ksteps / 16) ks = ksteps / 16;
  if (ks > 32) ks = 32;
  const long long cap_ws = (long long)(p.ws_bytes / ((size_t)p.M * p.Cout * sizeof(float)));
  if (ks > cap_ws) ks = cap_ws;
  if (ks < 1) return -1;   // no workspace for even one slab: the caller falls back to the tiled kernel
  p.ks_per = ceil_div(ksteps, ks);
  p.ksplit = ceil_div(ksteps, p.ks_per);
  p.partial = (float*)p.ws;
  if (p.dry) {
    g_dry_bytes = (size_t)p.ksplit * p.M * p.Cout * sizeof(float);
    return 0;
  }
  constexpr int LDS = 3 * 16 * WN * 128;
  TRACE_LDS_OPT_IN((&wsgemm_bf16_kernel<WN>), LDS, "igemm");
  hipLaunchKernelGGL((wsgemm_bf16_kernel<WN>), dim3(tiles_n, p.ksplit), dim3(256), LDS, s, p);
  // the reduce kernel sums `ksplit` slabs (one included) and applies the epilogue
  long long n4 = (long long)p.M * (p.Cout >> 2);
  hipLaunchKernelGGL(igemm_splitk_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, p);
  return check_launch("wsgemm_bf16");
}

// ---------------------------------------------------------------------------------------------
// Temporal 3x1x1 conv (stride 1, pad 1; conv1 of every inflated Bottleneck3d, lib/i3d/resnet3d.py:190-198,
// 19 % of the backbone's MACs) with the input tile shared by the three taps.  Through the generic loop each
// tap is its own K-step and reloads the (shifted) activation tile: 3 x (A + B) per 64 input channels.
// Here a tile is ALL T frames x Q = 128/T spatial positions, ordered (t, q): the tap dt of position (t, q) is
// LDS row (t + dt - 1) * Q + q, i.e. the same stage read Q rows lower / higher - an IMMEDIATE offset, and
// Q % 8 == 0 keeps the XOR swizzle phase.  Q zero rows sit in front of and behind the 128 tile rows (t = -1
// and t = T).  Per 64 input channels: one activation load + three weight loads (A through a 2-stage ring).
// ---------------------------------------------------------------------------------------------
template <int WM, int LOGQ>
__global__ __launch_bounds__(256) void tconv3_bf16_kernel(const IgemmParams p) {
  constexpr int WN = 4;
  constexpr int TC = 32 * WM;
  constexpr int Q = 1 << LOGQ;
  constexpr int A_BYTES = TC * 128;
  constexpr int B_BYTES = (128 + 2 * Q) * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const bsm = smem;                   // 2 activation stages (one per 64-channel chunk)
  char* const asm_ = smem + 2 * B_BYTES;    // 2 weight stages (one per K-step = (chunk, tap))

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fq = lane >> 4;
  const char* zero = reinterpret_cast<const char*>(g_zero_page);

  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x;
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
  }
  const int tile_n = bid % p.tiles_n;       // cout tile fastest: neighbours share the activation tile in L2
  const int rest = bid / p.tiles_n;
  const int tile_q = rest % p.tiles_m;
  const int n = rest / p.tiles_m;
  const int cout0 = tile_n * TC;
  const int hw0 = tile_q * Q;
  const int HW = p.Ho * p.Wo;
  const int T = p.To;
  const int Cin = p.Crun;

  // zero rows: [0, Q) and [Q + 128, 2Q + 128) of both activation stages
  for (int i = tid; i < 2 * 2 * Q * 8; i += 256) {
    const int st = i / (2 * Q * 8), r2 = i % (2 * Q * 8);
    const int row = r2 >> 3, ch = r2 & 7;
    const int lrow2 = row < Q ? row : row + 128;
    *reinterpret_cast<uint4*>(bsm + st * B_BYTES + lrow2 * 128 + ch * 16) = make_uint4(0, 0, 0, 0);
  }
  __syncthreads();

  const int lrow = wave * 8 + (lane >> 3);
  const int kseg = (lane & 7) ^ (lane >> 3);
  const char* ap[WM];
#pragma unroll
  for (int i = 0; i < WM; ++i) {
    const int row = min(cout0 + lrow + 32 * i, p.Cout - 1);
    ap[i] = reinterpret_cast<const char*>(p.w + (long long)row * p.ldw + a_kseg(wave, lane) * 8);
  }
  const char* bp[WN];
#pragma unroll
  for (int i = 0; i < WN; ++i) {
    const int pos = lrow + 32 * i;
    const int t = pos >> LOGQ, hw = hw0 + (pos & (Q - 1));
    bp[i] = hw < HW ? reinterpret_cast<const char*>(p.in + (((long long)n * T + t) * HW + hw) * p.pix + kseg * 8) : zero;
  }
  const long long a_tap = 2LL * Cin, a_chunk = 2LL * BK - 4LL * Cin;   // byte steps: next tap / first tap of the next chunk
  const int nchunks = Cin / BK;
  const int steps = 3 * nchunks;

  auto issue_a = [&](int stage, int dt) {
    char* a = asm_ + stage * A_BYTES + wave * 1024;
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      glds16(ap[i], a + i * 4096);
      ap[i] += dt == 2 ? a_chunk : a_tap;
    }
  };
  auto issue_b = [&](int stage) {
    char* b = bsm + stage * B_BYTES + Q * 128 + wave * 1024;
#pragma unroll
    for (int i = 0; i < WN; ++i) {
      glds16(bp[i], b + i * 4096);
      bp[i] += 2 * BK;
    }
  };

  f32x4 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int aoffs[2][2], boffs[2];
#pragma unroll
  for (int odd = 0; odd < 2; ++odd)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) aoffs[odd][s2] = lds_off_a(wm * (16 * WM) + perm_row(odd, fr), s2 * 4 + fq);
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) boffs[s2] = lds_off(wn * 64 + fr, s2 * 4 + fq);

  issue_b(0);
  issue_a(0, 0);
  int s = 0;
  // k6 = s mod 6 (compile-time in the callers): tap k6 % 3, weight stage k6 & 1, activation stage (k6 / 3) & 1
  auto step = [&](int k6) {
    const int dt = k6 % 3;
    wait_vmcnt<0>();
    ring_barrier();
    if (s + 1 < steps) issue_a((k6 + 1) & 1, (k6 + 1) % 3);
    if (dt == 0 && s + 3 < steps) issue_b(((k6 / 3) + 1) & 1);
    TRACE_SETTLE();
    const char* a = asm_ + (k6 & 1) * A_BYTES;
    const char* b = bsm + ((k6 / 3) & 1) * B_BYTES + dt * (Q * 128);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      bf16x8 af[WM], bfr[WN];
#pragma unroll
      for (int i = 0; i < WM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(a + aoffs[i & 1][s2] + (i >> 1) * (32 * 128));
#pragma unroll
      for (int j = 0; j < WN; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(b + boffs[s2] + j * (16 * 128));
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    ++s;
  };
  for (; s + 6 <= steps;) {
    step(0); step(1); step(2); step(3); step(4); step(5);
  }
  if (s < steps) {   // an odd number of chunks: one more chunk on activation stage 0
    step(0); step(1); step(2);
  }

  int mrow[WN];
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    const int pos = wn * 64 + j * 16 + fr;
    const int t = pos >> LOGQ, hw = hw0 + (pos & (Q - 1));
    mrow[j] = hw < HW ? (n * T + t) * HW + hw : -1;
  }
  igemm_epilogue_rows<WM, WN>(p, acc, cout0, mrow, wm, fq, 0);
}

template <int WM, int LOGQ>
static int launch_tconv3(IgemmParams& p, int N, hipStream_t s) {
  constexpr int TC = 32 * WM, Q = 1 << LOGQ;
  constexpr int LDS = 2 * (128 + 2 * Q) * 128 + 2 * TC * 128;
  p.tiles_n = ceil_div(p.Cout, TC);
  p.tiles_m = ceil_div(p.Ho * p.Wo, Q);
  p.ksplit = 1;
  p.ks_per = 0;
  const long long nwg = (long long)p.tiles_n * p.tiles_m * N;
  if (nwg > 0x7fffffffLL) {
    set_error("conv3d: grid too large");
    return 1;
  }
  if (p.dry) return 0;   // no scratch
  TRACE_LDS_OPT_IN((&tconv3_bf16_kernel<WM, LOGQ>), LDS, "igemm");
  hipLaunchKernelGGL((tconv3_bf16_kernel<WM, LOGQ>), dim3((unsigned)nwg), dim3(256), LDS, s, p);
  return check_launch("conv3d_tconv3");
}

// ---------------------------------------------------------------------------------------------
// Spatial 1x3x3 conv (stride 1, pad = dilation 1 or 2; conv2 of every Bottleneck3d, lib/i3d/resnet3d.py:199-207,
// 43 % of the backbone's MACs) with the input patch shared by the nine taps.  Through the generic loop every
// tap is its own K-step and re-fetches the (shifted) activation tile from L2: 9 x (A + B) per 64 input channels,
// and the layers are operand-delivery bound (L2 -> LDS), not MFMA bound.  Here a tile is TH = 2 WN output rows x
// 16 output columns of ONE frame and 64 couts; per 64-channel chunk its halo patch ((TH + 2D) rows x (16 + 2D)
// pixels, row pitch 24 pixels, one pixel = one 128-byte swizzled LDS row) is loaded ONCE and tap (dy, dx) of
// fragment j (= output row j of the wave's half) is patch-row fragment (j + dy D) - so the WN + 2D row fragments of a
// column dx are read ONCE (immediate offsets) and serve its three dy taps from registers - taken dx D pixels to the
// right - one of three per-lane base registers (the pitch is a multiple of 8 pixels, so only dx changes the XOR
// swizzle phase).  Per chunk: one patch load + nine 8 KB weight loads
// (4-stage ring, three K-steps in flight).  The patch of chunk c+1 is fetched in slices of SL loads per lane behind
// the weight loads of taps 0..5 of chunk c; the per-step wait is a counted vmcnt (see the main loop).
// K order is (chunk, tap), not (tap, channel): same sum, different f32 rounding order than the generic loop.
// ---------------------------------------------------------------------------------------------
constexpr int SC_PW = 24;                        // patch row pitch (pixels), % 8 == 0
constexpr int SC_NA = 4;                         // weight ring depth
template <int WN, int D, int KG>
__global__ __launch_bounds__(256 * KG) void sconv9_bf16_kernel(const IgemmParams p) {
  // KG = 2: eight waves; waves 4..7 repeat the 2 x 2 wave tile on the SECOND 32-wide half of every K-step (wave
  // group g takes MFMA K block g) and the two partial tiles are added through LDS at the end.  Two waves per SIMD
  // instead of one when the grid gives a CU a single workgroup: one wave's LDS reads hide behind the other's MFMAs.
  constexpr int WM = 2;
  constexpr int NWAVE = 4 * KG;
  constexpr int NS2 = 2 / KG;                    // 32-wide K halves per wave and K-step
  constexpr int TH = 2 * WN;                     // output rows per tile
  constexpr int PR = TH + 2 * D;                 // patch rows
  constexpr int GROUPS = PR * SC_PW / 8;         // 8-pixel groups = 1 KiB LDS-DMA wave instructions
  constexpr int NI = (GROUPS + NWAVE - 1) / NWAVE;   // patch loads per lane per chunk
  constexpr int SL = (NI + 5) / 6;               // loads per lane per slice (taps 0..5)
  constexpr int AL = 8 / NWAVE;                  // weight loads per lane per K-step
  constexpr int PATCH_BYTES = NWAVE * NI * 1024;
  constexpr int A_BYTES = 64 * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const asm_ = smem;                       // SC_NA weight stages
  char* const bsm = smem + SC_NA * A_BYTES;      // 1 or 2 patch stages

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = wave >> 2, w4 = wave & 3;
  const int wm = w4 >> 1, wn = w4 & 1;
  const int fr = lane & 15, fq = lane >> 4;
  const char* zero = reinterpret_cast<const char*>(g_zero_page);

  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x;
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
  }
  const int H = p.Ho, W = p.Wo;
  const int tiles_x = (W + 15) >> 4, tiles_y = (H + TH - 1) / TH;
  const int tile_n = bid % p.tiles_n;            // cout tile fastest: neighbours share the patch in L2
  int rest = bid / p.tiles_n;
  const int tx = rest % tiles_x;
  rest /= tiles_x;
  const int ty = rest % tiles_y;
  const int f = rest / tiles_y;                  // frame = n * To + t
  const int cout0 = tile_n * 64;
  const int y0 = ty * TH, x0 = tx * 16;
  const int Cin = p.Crun;

  const int kseg = (lane & 7) ^ (lane >> 3);
  const char* ap[AL];      // weight rows: 1 KiB group wave + NWAVE i = couts 8 (wave + NWAVE i) .. + 8
#pragma unroll
  for (int i = 0; i < AL; ++i) {
    const int row = min(cout0 + (wave + NWAVE * i) * 8 + (lane >> 3), p.Cout - 1);
    ap[i] = reinterpret_cast<const char*>(p.w + (long long)row * p.ldw + a_kseg(wave, lane) * 8);
  }
  const uint16_t* frame = p.in + (long long)(f / p.To) * p.in_n_stride + (long long)(f % p.To) * p.Hi * p.Wi * p.pix;
  const char* bp[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int pixel = (wave + NWAVE * i) * 8 + (lane >> 3);
    const int pr = pixel / SC_PW, pc = pixel - pr * SC_PW;
    const int y = y0 - D + pr, x = x0 - D + pc;
    const bool ok = pr < PR && pc < 16 + 2 * D && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
    bp[i] = ok ? reinterpret_cast<const char*>(frame + ((long long)y * W + x) * p.pix + kseg * 8) : zero;
  }
  const long long a_tap = 2LL * Cin, a_chunk = 2LL * BK - 16LL * Cin;   // byte steps: next tap / tap 0 of the next chunk
  const int nchunks = Cin / BK;

  // weights of K-step (chunk, pos): taps are visited column by column (pos = 3 dx + dy -> tap 3 dy + dx, so that the
  // three dy taps of a column reuse the same patch-row fragments from registers): +3, +3, -5 taps, ..., then tap 0 of
  // the next chunk
  int a_pos_next = 0;      // pos of the next K-step to issue (uniform)
  auto issue_a = [&](int stage) {
    char* a = asm_ + stage * A_BYTES + wave * 1024;
    const long long adv = a_pos_next == 8 ? a_chunk : (a_pos_next % 3 == 2 ? -5 * a_tap : 3 * a_tap);
#pragma unroll
    for (int i = 0; i < AL; ++i) {
      glds16(ap[i], a + i * (NWAVE * 1024));
      ap[i] += adv;
    }
    a_pos_next = a_pos_next == 8 ? 0 : a_pos_next + 1;
  };
  auto issue_patch = [&](int stage, int i) {     // load i of the lane: 8-pixel group wave + NWAVE i
    glds16(bp[i], bsm + stage * PATCH_BYTES + (wave + NWAVE * i) * 1024);
    bp[i] += 2 * BK;
  };

  f32x4 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int aoffs[2][NS2], boffs[3][NS2];
#pragma unroll
  for (int odd = 0; odd < 2; ++odd)
#pragma unroll
    for (int h = 0; h < NS2; ++h) aoffs[odd][h] = lds_off_a(wm * (16 * WM) + perm_row(odd, fr), (kg * NS2 + h) * 4 + fq);
#pragma unroll
  for (int dx = 0; dx < 3; ++dx)
#pragma unroll
    for (int h = 0; h < NS2; ++h) {
      const int px = fr + dx * D;
      boffs[dx][h] = (wn * WN * SC_PW + px) * 128 + ((((kg * NS2 + h) * 4 + fq) ^ (px & 7)) << 4);
    }

  // prologue: the whole patch of chunk 0, then the weights of the first SC_NA - 1 K-steps (a chunk has 9 >= 3)
#pragma unroll
  for (int i = 0; i < NI; ++i) issue_patch(0, i);
#pragma unroll
  for (int k = 0; k < SC_NA - 1; ++k) issue_a(k);

  // Loads retire in order, so "K-step v's weights have landed" = "at most the loads issued after them are still in
  // flight".  Step u issues [weights of step u+3][patch slice of tap(u)]; the weights of step v were issued first
  // in step v-3, hence the allowance slice(v-3) + AL + slice(v-2) + AL + slice(v-1).  Slices only exist on taps 0..5
  // of a chunk that has a successor, so at tap 0 the allowance (2 AL) already implies the whole next patch has landed.
  int astage = 0;          // ring stage of the current K-step (uniform)
  for (int c = 0; c < nchunks; ++c) {
    const bool steady = c + 1 < nchunks;          // a next chunk exists: its patch is fetched behind this chunk's weights
    const int pst = c & 1;
    const char* const bbase = bsm + pst * PATCH_BYTES;
    bf16x8 brow[WN + 2 * D][NS2];   // patch-row fragments of the current column dx
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {   // tap = position in the visiting order (3 dx + dy)
      const int dx = tap / 3, dy = tap % 3;
      auto cnt = [&](int t) { return (t < 0 || t > 5) ? 0 : (NI - t * SL <= 0 ? 0 : (NI - t * SL < SL ? NI - t * SL : SL)); };
      const int al_s = cnt(tap - 3) + cnt(tap - 2) + cnt(tap - 1) + 2 * AL;
      const int al_l = (tap <= 7 ? AL : 0) + (tap <= 6 ? AL : 0);
      switch (steady ? al_s : al_l) {     // two compile-time values after unrolling
        case 1: wait_vmcnt<1>(); break;
        case 2: wait_vmcnt<2>(); break;
        case 3: wait_vmcnt<3>(); break;
        case 4: wait_vmcnt<4>(); break;
        case 5: wait_vmcnt<5>(); break;
        case 6: wait_vmcnt<6>(); break;
        case 7: wait_vmcnt<7>(); break;
        case 8: wait_vmcnt<8>(); break;
        case 9: wait_vmcnt<9>(); break;
        case 10: wait_vmcnt<10>(); break;
        case 11: wait_vmcnt<11>(); break;
        case 12: wait_vmcnt<12>(); break;
        case 13: wait_vmcnt<13>(); break;
        default: wait_vmcnt<0>(); break;
      }
      ring_barrier();
      if (tap + SC_NA - 1 <= 8 || steady) issue_a((astage + SC_NA - 1) & (SC_NA - 1));
      if (steady) {
#pragma unroll
        for (int i = tap * SL; i < (tap + 1) * SL && i < NI && tap <= 5; ++i) issue_patch(pst ^ 1, i);
      }
      TRACE_SETTLE();
      const char* a = asm_ + astage * A_BYTES;
      if (dy == 0) {
#pragma unroll
        for (int h = 0; h < NS2; ++h)
#pragma unroll
          for (int r = 0; r < WN + 2 * D; ++r)
            brow[r][h] = *reinterpret_cast<const bf16x8*>(bbase + boffs[dx][h] + r * (SC_PW * 128));
      }
#pragma unroll
      for (int h = 0; h < NS2; ++h) {
        bf16x8 af[WM];
#pragma unroll
        for (int i = 0; i < WM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(a + aoffs[i & 1][h]);
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int j = 0; j < WN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], brow[j + dy * D][h], acc[i][j], 0, 0, 0);
      }
      astage = (astage + 1) & (SC_NA - 1);
    }
  }

  if (KG == 2) {   // add the second K half's tile: [wave tile][register][lane] floats over the (now idle) operand stages
    float* red = reinterpret_cast<float*>(smem) + w4 * (WM * WN * 4 * 64) + lane;
    __syncthreads();
    if (kg == 1) {
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) red[((i * WN + j) * 4 + r) * 64] = acc[i][j][r];
    }
    __syncthreads();
    if (kg == 1) return;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] += red[((i * WN + j) * 4 + r) * 64];
  }

  int mrow[WN];
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    const int y = y0 + wn * WN + j, x = x0 + fr;
    mrow[j] = (y < H && x < W) ? (f * H + y) * W + x : -1;
  }
  igemm_epilogue_rows<WM, WN>(p, acc, cout0, mrow, wm, fq, 0);
}

// TRACE_SCONV_KG=1|2 forces the wave-group count of sconv9 (A/B measurements); default: by workgroups per CU
static const int g_sconv_kg = getenv("TRACE_SCONV_KG") ? atoi(getenv("TRACE_SCONV_KG")) : 0;

template <int WN, int D, int KG>
static int launch_sconv9_kg(IgemmParams& p, long long nwg, hipStream_t s) {
  constexpr int NWAVE = 4 * KG;
  constexpr int NI = ((2 * WN + 2 * D) * SC_PW / 8 + NWAVE - 1) / NWAVE;
  constexpr int PATCH_BYTES = NWAVE * NI * 1024;
  constexpr int LDS_MAX = 2 * PATCH_BYTES + SC_NA * 64 * 128;
  static_assert(KG == 1 || PATCH_BYTES + SC_NA * 64 * 128 >= 4 * 2 * WN * 4 * 64 * 4, "reduction scratch must fit the operand stages");
  p.tiles_n = ceil_div(p.Cout, 64);
  p.ksplit = 1;
  p.ks_per = 0;
  if (p.dry) return 0;   // no scratch
  TRACE_LDS_OPT_IN((&sconv9_bf16_kernel<WN, D, KG>), LDS_MAX, "igemm");
  const int lds = (p.Crun > BK ? 2 : 1) * PATCH_BYTES + SC_NA * 64 * 128;   // one 64-channel chunk: no second patch stage
  hipLaunchKernelGGL((sconv9_bf16_kernel<WN, D, KG>), dim3((unsigned)nwg), dim3(256 * KG), lds, s, p);
  return check_launch("conv3d_sconv9");
}

template <int WN, int D>
static int launch_sconv9(IgemmParams& p, long long nwg, hipStream_t s) {
  // one workgroup per CU (by grid size or by its LDS footprint): eight waves, else four
  const int kg = g_sconv_kg ? g_sconv_kg : (nwg <= 320 || p.Crun > BK ? 2 : 1);
  return kg == 2 ? launch_sconv9_kg<WN, D, 2>(p, nwg, s) : launch_sconv9_kg<WN, D, 1>(p, nwg, s);
}

// Launch configuration: tile (cout x positions), LDS ring depth, split-K factor.  0 = choose here.
struct IgemmTune {
  int tile;    // 1: 64x64, 2: 64c x 128p, 3: 128c x 64p, 4: 128x128, 5: 128c x 256p, 6: 256c x 128p, 7: 256x256,
               // 8: 128c x 192p, 9: 256c x 192p, 10: register-fed weight streaming, 12: 128c x 160p (see launch_igemm)
  int nst;     // 2..4
  int ksplit;  // >= 1
};
static thread_local IgemmTune g_tune = {0, 0, 0};
// TRACE_IGEMM_V2=1 keeps the v2 main loop (A/B measurements)
static const bool g_force_v2 = getenv("TRACE_IGEMM_V2") != nullptr;
// TRACE_NO_TCONV=1 sends temporal convs through the generic loop (A/B measurements)
static const bool g_no_tconv = getenv("TRACE_NO_TCONV") != nullptr;
// TRACE_NO_SCONV=1 sends 1x3x3 convs through the generic loop (A/B measurements)
static const bool g_no_sconv = getenv("TRACE_NO_SCONV") != nullptr;
// TRACE_NO_SMALLK=1 keeps the 128 x 128 tiles for dense problems of one or two K-steps (A/B measurements)
static const bool g_no_smallk = getenv("TRACE_NO_SMALLK") != nullptr;

template <int WM, int WN>
static int launch_cfg(IgemmParams& p, int batch, hipStream_t s, int nst_req, int ksplit_req) {
  constexpr int TC = 32 * WM, TP = 32 * WN;
  p.tiles_n = ceil_div(p.Cout, TC);
  p.tiles_m = ceil_div(p.M, TP);
  long long nblk = (long long)p.tiles_m * p.tiles_n;
  if (nblk > 0x7fffffffLL) {
    set_error("igemm: grid too large");
    return 1;
  }
  const int ksteps = ceil_div(p.Ktot, BK);
  p.m_fast = (p.tiles_m <= 4 && ksteps >= 64) ? 1 : 0;
  // split-K when the tile grid cannot fill 256 CUs and K is deep (weight-streaming FC layers,
  // single-clip convs): slices write f32 slabs to the caller's workspace, a second launch reduces.
  p.ksplit = 1;
  p.ks_per = ksteps;
  if (batch == 1 && p.ws) {
    long long ks = 1;
    const long long cap_ws = (long long)(p.ws_bytes / ((size_t)p.M * p.Cout * sizeof(float)));
    if (ksplit_req > 0) {
      ks = ksplit_req;
    } else if (nblk < 256 && ksteps >= 128) {
      // deep K (weight-streaming FCs, the 12544-wide pair FC): one wave of workgroups (~2 per CU with small
      // tiles, 1 per CU with the big ones), >= 16 K-steps per slice; measured on cold operands
      // (round-1 sweep): 21 slices for the 160 x 3072 x 100352 FC, 4 for 3040 x 256 x 12544
      const long long target = (TC + TP >= 256) ? 256 : 512;
      ks = (target + nblk / 2) / nblk;
      if (TC + TP >= 256 && p.M <= 256 && ks < 21 && ksteps >= 1024) ks = 21;
      if (ks > ksteps / 16) ks = ksteps / 16;
      if (ks > 32) ks = 32;
      // few cout tiles (<= 8: fc_r, 168 x 1024 x 100352, 90 -> 68 us): a multiple of 8 slices, pinned to XCDs (see the
      // kernel) so that the activation K-slice is fetched into one L2 only; with many cout tiles (24 for the
      // 3072-wide FC) an XCD already shares the slice between its tiles and whole rounds of workgroups matter more
      if (ks >= 12 && nblk <= 8) ks = (ks + 4) / 8 * 8;
    } else if (nblk < 128 && ksteps >= 32) {
      // badly under-filled grids (single-clip convs): a few slices of >= 16 K-steps each
      ks = 256 / nblk;
      if (ks > ksteps / 16) ks = ksteps / 16;
    }
    if (ks < 1) ks = 1;
    if (ks > ksteps) ks = ksteps;
    if (ks > cap_ws) ks = cap_ws;
    if (ks > 1) {
      p.ks_per = ceil_div(ksteps, ks);
      p.ksplit = ceil_div(ksteps, p.ks_per);
      p.partial = (float*)p.ws;
    }
  }
  if (p.dry) {
    g_dry_bytes = p.ksplit > 1 ? (size_t)p.ksplit * p.M * p.Cout * sizeof(float) : 0;
    return 0;
  }
  p.xcd_slices = (p.ksplit >= 8 && p.ksplit % 8 == 0 && batch == 1) ? 1 : 0;
  // weight streaming (pre-tiled weights, all positions in one tile): each weight byte is fetched once - non-temporal
  static const bool no_nt = getenv("TRACE_NO_WSTREAM_NT") != nullptr;   // A/B measurements
  p.w_stream = (p.w_tiled && p.tiles_m == 1 && batch == 1 && !no_nt) ? 1 : 0;
  const dim3 grid((unsigned)nblk, p.ksplit, batch);
  // ring depth.  v2 main loop: 2 stages (most workgroups per CU).  v3: the leaner K loop is latency-bound
  // when few workgroups share a CU, so the ring goes to 3 (4 for single-workgroup-per-CU grids) whenever
  // every workgroup of the launch still fits on the chip at once with the deeper ring.
  const size_t stage = (size_t)(TC + TP) * 128;
  const bool v3 = !g_force_v2 && (p.dense || (p.Crun % BK == 0 && p.Crun <= 2048 && p.kt * p.kh * p.kw <= 32));
  int nst = nst_req ? nst_req : 2;
  if (!nst_req && v3) {
    const long long total = nblk * p.ksplit * batch;
    const long long per_cu = (total + 255) / 256;
    const int ks_slice = p.ks_per;
    if (per_cu == 1 && 4 * stage <= 160 * 1024 && ks_slice >= 8 && (TC + TP) <= 192) nst = 4;
    else if (per_cu * 3 * stage <= 160 * 1024 && ks_slice >= 4) nst = 3;
    else if (ksteps >= 1024 && p.M <= 256 && 3 * stage <= 160 * 1024) nst = 3;   // weight streaming: bytes in flight first
  }
  if (nst < 2) nst = 2;
  if (nst > 4) nst = 4;
  while (nst > 2 && nst * stage > 160 * 1024) --nst;  // 160 KiB of LDS per CU
  // a K slice shorter than the ring touches only its first ks_per stages (the loops never issue past the slice):
  // allocate just those, so that single-K-step layers (64 -> 256 1x1x1 at 200704 positions) get 4 workgroups per CU
  const int lds_stages = (v3 && p.ks_per < nst) ? p.ks_per : nst;
  {
#define TRACE_IGEMM2(D, S)                                                                               \
  do {                                                                                                   \
    TRACE_LDS_OPT_IN((&igemm2_bf16_kernel<WM, WN, D, S>), S * stage, "igemm"); \
    hipLaunchKernelGGL((igemm2_bf16_kernel<WM, WN, D, S>), grid, dim3(256), S * stage, s, p);           \
  } while (0)
#define TRACE_IGEMM3(D, S)                                                                               \
  do {                                                                                                   \
    TRACE_LDS_OPT_IN((&igemm3_bf16_kernel<WM, WN, D, S>), S * stage, "igemm"); \
    hipLaunchKernelGGL((igemm3_bf16_kernel<WM, WN, D, S>), grid, dim3(256), lds_stages * stage, s, p);  \
  } while (0)
    // v3 main loop (eligibility computed above): a K-step must not straddle a tap, taps must fit the 32-bit
    // validity mask, a halo pointer must stay inside the zero page while it walks a tap
    if (v3) {
      if (p.dense) {
        if (nst == 2) TRACE_IGEMM3(true, 2); else if (nst == 3) TRACE_IGEMM3(true, 3); else TRACE_IGEMM3(true, 4);
      } else {
        if (nst == 2) TRACE_IGEMM3(false, 2); else if (nst == 3) TRACE_IGEMM3(false, 3); else TRACE_IGEMM3(false, 4);
      }
    } else if (p.dense) {
      if (nst == 2) TRACE_IGEMM2(true, 2); else if (nst == 3) TRACE_IGEMM2(true, 3); else TRACE_IGEMM2(true, 4);
    } else {
      if (nst == 2) TRACE_IGEMM2(false, 2); else if (nst == 3) TRACE_IGEMM2(false, 3); else TRACE_IGEMM2(false, 4);
    }
#undef TRACE_IGEMM2
#undef TRACE_IGEMM3
  }
  if (p.ksplit > 1) {
    long long n4 = (long long)p.M * (p.Cout >> 2);
    hipLaunchKernelGGL(igemm_splitk_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, p);
  }
  return check_launch("igemm_bf16");
}

static int launch_igemm(IgemmParams& p, int batch, hipStream_t s) {
  if (p.M <= 0 || p.Cout <= 0 || batch <= 0) return 0;
  p.narrow = (p.Cout % 8 != 0 || p.ldo % 8 != 0) ? 1 : 0;
  p.dense = (p.kt == 1 && p.kh == 1 && p.kw == 1 && p.st == 1 && p.sh == 1 && p.sw == 1 && p.pt == 0 && p.ph == 0 &&
             p.pw == 0 && p.Crun == p.Ktot && p.in_n_stride == (long long)p.Ti * p.Hi * p.Wi * p.pix)
                ? 1 : 0;
  const IgemmTune t = g_tune;
  int tile = t.tile;
  // temporal 3x1x1 convs: the tile spans all T frames and is shared by the three taps (tconv3_bf16_kernel)
  if (!tile && !g_force_v2 && !g_no_tconv && batch == 1 && !p.narrow && p.kt == 3 && p.kh == 1 && p.kw == 1 && p.st == 1 && p.sh == 1 &&
      p.sw == 1 && p.pt == 1 && p.ph == 0 && p.pw == 0 && p.dt == 1 && p.Crun % BK == 0 && p.Crun <= 2048 &&
      (p.To == 2 || p.To == 4 || p.To == 8) && p.Ti == p.To && p.M < 0x40000000) {
    const int N = p.M / (p.To * p.Ho * p.Wo);
    const int Q = 128 / p.To;
    const long long wg4 = (long long)ceil_div(p.Cout, 128) * ceil_div(p.Ho * p.Wo, Q) * N;
    const bool small = p.Cout <= 64 || wg4 < 384;    // 64-cout tiles when 128-cout tiles cannot fill the chip
    const long long wg = small ? (long long)ceil_div(p.Cout, 64) * ceil_div(p.Ho * p.Wo, Q) * N : wg4;
    if (wg >= 300)   // measured: below ~1 workgroup per CU the generic loop's deeper ring wins (1024->256 @ 4x14x14)
    switch (p.To) {
      case 8: return small ? launch_tconv3<2, 4>(p, N, s) : launch_tconv3<4, 4>(p, N, s);
      case 4: return small ? launch_tconv3<2, 5>(p, N, s) : launch_tconv3<4, 5>(p, N, s);
      default: return small ? launch_tconv3<2, 6>(p, N, s) : launch_tconv3<4, 6>(p, N, s);
    }
  }
  // spatial 1x3x3 convs (stride 1, pad = dilation): the halo patch is shared by the nine taps (sconv9_bf16_kernel)
  if (!tile && !g_force_v2 && !g_no_sconv && batch == 1 && !p.narrow && p.kt == 1 && p.kh == 3 && p.kw == 3 && p.st == 1 &&
      p.sh == 1 && p.sw == 1 && p.pt == 0 && p.ph == p.dh && p.pw == p.dw && p.dh == p.dw && (p.dh == 1 || p.dh == 2) &&
      p.Crun % BK == 0 && p.Crun <= 2048 && p.Ktot == 9 * p.Crun && p.Ho == p.Hi && p.Wo == p.Wi && p.To == p.Ti &&
      p.M < 0x40000000 && p.M % (p.Ho * p.Wo) == 0) {
    const long long F = p.M / (p.Ho * p.Wo);
    const long long per_frame = (long long)ceil_div(p.Wo, 16) * ceil_div(p.Cout, 64);
    const long long wg14 = F * ceil_div(p.Ho, 14) * per_frame, wg8 = F * ceil_div(p.Ho, 8) * per_frame;
    // 14-row tiles (fewest weight re-reads) when they still give every CU a workgroup, else 8-row tiles
    if (wg14 >= 200) return p.dh == 1 ? launch_sconv9<7, 1>(p, wg14, s) : launch_sconv9<7, 2>(p, wg14, s);
    if (wg8 >= 128) return p.dh == 1 ? launch_sconv9<4, 1>(p, wg8, s) : launch_sconv9<4, 2>(p, wg8, s);
  }
  if (p.in2) p.ws = nullptr;   // two activation sources: the switch K-step is counted from K-step 0 (no split-K)
  if (p.pool_t2) {   // checked by the entry point: dense, bf16 out, even To; 128 x 128 tiles, no split-K
    p.ws = nullptr;
    return launch_cfg<4, 4>(p, batch, s, t.nst, 1);
  }
  if (p.w_tiled) {   // pre-tiled weights: 128-cout tiles of the v3 loop only
    if (!p.dense || p.Cout % 128 != 0 || p.Ktot % BK != 0 || g_force_v2) {
      set_error("gemm: TRACE_EPI_WTILED needs a dense problem with N %% 128 == 0 and K %% 64 == 0");
      return 1;
    }
    if (tile != 3 && tile != 4 && tile != 8 && tile != 12) tile = p.M <= 64 ? 3 : (p.M <= 128 || p.M > 192 ? 4 : (p.M <= 160 ? 12 : 8));
    switch (tile) {
      case 3: return launch_cfg<4, 2>(p, batch, s, t.nst, t.ksplit);
      case 8: return launch_cfg<4, 6>(p, batch, s, t.nst, t.ksplit);
      case 12: return launch_cfg<4, 5>(p, batch, s, t.nst, t.ksplit);
      default: return launch_cfg<4, 4>(p, batch, s, t.nst, t.ksplit);
    }
  }
  // register-fed weight-streaming kernel: selectable (tile id 10) but not the default - measured 198 us vs 182 us for the
  // tiled loop on the 160 x 3072 x 100352 FC (both move the same bytes; the tiled loop with pre-tiled weights wins)
  if (tile == 10 && p.dense && batch == 1 && p.ws && !p.narrow && p.M <= 192 && p.Ktot % BK == 0 &&
      p.Ktot / BK >= 512 && p.Cout % 4 == 0) {
    int rc = p.M <= 64 ? launch_wsgemm<4>(p, s, t.ksplit) : (p.M <= 128 ? launch_wsgemm<8>(p, s, t.ksplit) : launch_wsgemm<12>(p, s, t.ksplit));
    if (rc >= 0) return rc;
  }
  if (tile == 10) tile = 0;
  if (!tile) {
    // Tile choice: keep >= ~2 workgroups per CU in flight when the problem allows.
    auto blocks = [&](int tc, int tp) { return (long long)ceil_div(p.Cout, tc) * ceil_div(p.M, tp) * batch; };
    const int ksteps = ceil_div(p.Ktot, BK);
    if (p.Cout <= 64) tile = 2;                                                   // 64 couts x 128 positions
    else if (p.ws && batch == 1 && p.M <= 192 && ksteps >= 1024)                  // weight streaming + split-K:
      tile = p.M <= 64 ? 3 : (p.M <= 128 ? 4 : (p.M <= 160 ? 12 : 8));            //   all positions in one tile
    else if (p.ws && batch == 1 && ksteps >= 128 && blocks(128, 128) <= 128) tile = 4;   // deep K: big tiles + split-K
    else if (!g_no_smallk && ksteps <= 2 && blocks(64, 64) >= 1024) {
      // one or two K-steps (layer1 / layer2 conv3 at the bench batch: 200704 x 256 x 64, 25088 x 512 x 128): the launch
      // is HBM traffic plus store bursts, and small tiles keep more workgroups' loads and stores in flight per CU -
      // measured 53.0 -> 41.5 us (K = 64, + residual), 38.0 -> 31.1 (K = 64), 18.4 -> 16.9 (K = 128, + residual)
      tile = ksteps == 1 ? 3 : 1;
    }
    else if (blocks(128, 128) >= 384) {
      // whole rounds of workgroups matter for mid-sized grids: 128x128 tiles (64 KB, 2 per CU = 512 slots) against
      // 128c x 64p (48 KB, 3 per CU = 768 slots), cost = rounds x operand bytes per K-step.  Cold sweep
      // (round-1 sweep): 6272 x 1536 x 1024 (588 tiles = 2 rounds of 512) 34.5 -> 29.9 us with the smaller
      // tile; every other backbone shape keeps 128x128 under the same rule
      tile = 4;
      const long long b4 = blocks(128, 128), b3 = blocks(128, 64);
      if (ksteps >= 8 && b4 <= 1536 && ((b3 + 767) / 768) * 192 < ((b4 + 511) / 512) * 256) tile = 3;
    }
    else if (blocks(128, 64) >= 384) tile = 3;
    else tile = 1;                                                                // 64 x 64: small problems
  }
  switch (tile) {
    case 1: return launch_cfg<2, 2>(p, batch, s, t.nst, t.ksplit);
    case 2: return launch_cfg<2, 4>(p, batch, s, t.nst, t.ksplit);
    case 3: return launch_cfg<4, 2>(p, batch, s, t.nst, t.ksplit);
    case 5: return launch_cfg<4, 8>(p, batch, s, t.nst, t.ksplit);
    case 6: return launch_cfg<8, 4>(p, batch, s, t.nst, t.ksplit);
    case 7: return launch_cfg<8, 8>(p, batch, s, t.nst, t.ksplit);
    case 8: return launch_cfg<4, 6>(p, batch, s, t.nst, t.ksplit);
    case 9: return launch_cfg<8, 6>(p, batch, s, t.nst, t.ksplit);
    case 12: return launch_cfg<4, 5>(p, batch, s, t.nst, t.ksplit);
    default: return launch_cfg<4, 4>(p, batch, s, t.nst, t.ksplit);
  }
}

}  // namespace trace

using namespace trace;

extern "C" int trace_igemm_tune(int tile, int nst, int ksplit) {
  TRACE_REQUIRE(tile >= 0 && tile <= 12 && tile != 11 && nst >= 0 && nst <= 4 && nst != 1 && ksplit >= 0, "igemm_tune: bad configuration");
  g_tune = IgemmTune{tile, nst, ksplit};
  return 0;
}

static int conv3d_impl(const uint16_t* in, int N, int Ti, int Hi, int Wi, int Cin,
                       const uint16_t* w, const float* bias, int Cout, int kt,
                       int kh, int kw, int st, int sh, int sw, int pt, int ph,
                       int pw, int dt, int dh, int dw, const uint16_t* residual,
                       int epilogue, void* out, void* workspace,
                       size_t workspace_bytes, trace_stream_t stream, int dry) {
  TRACE_REQUIRE(N >= 0 && Ti > 0 && Hi > 0 && Wi > 0 && Cin > 0 && Cout > 0, "conv3d: bad shape");
  TRACE_REQUIRE(Cin % 8 == 0, "conv3d: Cin=%d must be a multiple of 8", Cin);
  TRACE_REQUIRE(Cout % 8 == 0, "conv3d: Cout=%d must be a multiple of 8", Cout);
  TRACE_REQUIRE(kt > 0 && kh > 0 && kw > 0 && st > 0 && sh > 0 && sw > 0 && dt > 0 && dh > 0 && dw > 0 &&
                    pt >= 0 && ph >= 0 && pw >= 0,
                "conv3d: bad kernel/stride/padding/dilation");
  const int To = (Ti + 2 * pt - dt * (kt - 1) - 1) / st + 1;
  const int Ho = (Hi + 2 * ph - dh * (kh - 1) - 1) / sh + 1;
  const int Wo = (Wi + 2 * pw - dw * (kw - 1) - 1) / sw + 1;
  TRACE_REQUIRE(To > 0 && Ho > 0 && Wo > 0, "conv3d: empty output (%d,%d,%d)", To, Ho, Wo);
  long long M = (long long)N * To * Ho * Wo;
  TRACE_REQUIRE(M < 0x7fffffffLL, "conv3d: too many output positions");
  if (M == 0) return 0;
  TRACE_REQUIRE(in && w && out, "conv3d: null pointer");
  TRACE_REQUIRE(!(epilogue & TRACE_EPI_RESIDUAL) || residual, "conv3d: residual flag without pointer");
  TRACE_REQUIRE((((uintptr_t)in | (uintptr_t)w | (uintptr_t)out | (uintptr_t)residual | (uintptr_t)bias) & 15) == 0,
                "conv3d: pointers must be 16-byte aligned");
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  p.in = in;
  p.w = w;
  p.bias = bias;
  p.res = residual;
  p.out = out;
  p.M = (int)M;
  p.Cout = Cout;
  p.Crun = Cin;
  p.Ktot = kt * kh * kw * Cin;
  p.To = To; p.Ho = Ho; p.Wo = Wo; p.Ti = Ti; p.Hi = Hi; p.Wi = Wi;
  p.kt = kt; p.kh = kh; p.kw = kw;
  p.st = st; p.sh = sh; p.sw = sw; p.pt = pt; p.ph = ph; p.pw = pw; p.dt = dt; p.dh = dh; p.dw = dw;
  p.in_n_stride = (long long)Ti * Hi * Wi * Cin;
  p.pix = Cin;
  p.ldw = p.Ktot;
  p.ldo = Cout;
  p.alpha = 1.0f;
  p.epi = epilogue & ~TRACE_EPI_POOL_T2;
  if (epilogue & TRACE_EPI_POOL_T2) {
    TRACE_REQUIRE(kt == 1 && kh == 1 && kw == 1 && st == 1 && sh == 1 && sw == 1 && pt == 0 && ph == 0 && pw == 0,
                  "conv3d: TRACE_EPI_POOL_T2 needs a 1x1x1 stride-1 conv");
    TRACE_REQUIRE(To % 2 == 0 && !(epilogue & (TRACE_EPI_OUT_F32 | TRACE_EPI_LEAKY | TRACE_EPI_RES_POST)) && !g_force_v2,
                  "conv3d: TRACE_EPI_POOL_T2 needs an even frame count and a bf16 ReLU / linear epilogue");
    p.pool_t2 = 1;
  }
  p.ws = ((uintptr_t)workspace & 15) ? nullptr : workspace;
  p.ws_bytes = workspace_bytes;
  p.dry = dry;
  return launch_igemm(p, 1, (hipStream_t)stream);
}

extern "C" int trace_conv3d_igemm_bf16(const uint16_t* in, int N, int Ti, int Hi, int Wi, int Cin,
                                       const uint16_t* w, const float* bias, int Cout, int kt,
                                       int kh, int kw, int st, int sh, int sw, int pt, int ph,
                                       int pw, int dt, int dh, int dw, const uint16_t* residual,
                                       int epilogue, void* out, void* workspace,
                                       size_t workspace_bytes, trace_stream_t stream) {
  return conv3d_impl(in, N, Ti, Hi, Wi, Cin, w, bias, Cout, kt, kh, kw, st, sh, sw, pt, ph, pw, dt, dh, dw, residual, epilogue,
                     out, workspace, workspace_bytes, stream, 0);
}

// The split-K scratch the launch rules WANT for this problem: with at least this much workspace the plan (tile, K slices,
// hence the f32 summation order) does not depend on the caller's buffer; with less, fewer slices are used (or none).
extern "C" size_t trace_conv3d_workspace_bytes(int N, int Ti, int Hi, int Wi, int Cin, int Cout, int kt, int kh, int kw,
                                               int st, int sh, int sw, int pt, int ph, int pw, int dt, int dh, int dw,
                                               int epilogue) {
  static const uint4 dummy = {};
  const uint16_t* a = reinterpret_cast<const uint16_t*>(&dummy);
  g_dry_bytes = 0;
  const int rc = conv3d_impl(a, N, Ti, Hi, Wi, Cin, a, nullptr, Cout, kt, kh, kw, st, sh, sw, pt, ph, pw, dt, dh, dw,
                             (epilogue & TRACE_EPI_RESIDUAL) ? a : nullptr, epilogue, (void*)a, (void*)a, (size_t)-1, nullptr, 1);
  return rc ? 0 : g_dry_bytes;
}

extern "C" int trace_conv3d_stem_bf16(const uint16_t* in_padded, int N, int Tp, int Hp, int Wp, int Cp,
                                      const uint16_t* w, const float* bias, int Cout, int kt, int kh,
                                      int kwp, int st, int sh, int sw, int To, int Ho, int Wo,
                                      int epilogue, void* out, trace_stream_t stream) {
  TRACE_REQUIRE(N >= 0 && Tp > 0 && Hp > 0 && Wp > 0 && Cp > 0 && Cout > 0, "conv3d_stem: bad shape");
  TRACE_REQUIRE((kwp * Cp) % 8 == 0 && (sw * Cp) % 8 == 0 && (Wp * Cp) % 8 == 0,
                "conv3d_stem: kwp*Cp, sw*Cp and Wp*Cp must be multiples of 8 (16-byte pieces)");
  TRACE_REQUIRE(Cout % 8 == 0, "conv3d_stem: Cout must be a multiple of 8");
  TRACE_REQUIRE((To - 1) * st + kt <= Tp && (Ho - 1) * sh + kh <= Hp && (Wo - 1) * sw + kwp <= Wp,
                "conv3d_stem: padded input (%d,%d,%d) too small for output (%d,%d,%d)", Tp, Hp, Wp, To, Ho, Wo);
  long long M = (long long)N * To * Ho * Wo;
  TRACE_REQUIRE(M < 0x7fffffffLL, "conv3d_stem: too many output positions");
  if (M == 0) return 0;
  TRACE_REQUIRE(in_padded && w && out, "conv3d_stem: null pointer");
  TRACE_REQUIRE((((uintptr_t)in_padded | (uintptr_t)w | (uintptr_t)out | (uintptr_t)bias) & 15) == 0,
                "conv3d_stem: pointers must be 16-byte aligned");
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  p.in = in_padded;
  p.w = w;
  p.bias = bias;
  p.out = out;
  p.M = (int)M;
  p.Cout = Cout;
  p.Crun = kwp * Cp;  // one (kt,kh) tap = a contiguous run of kwp pixels x Cp channels
  p.Ktot = kt * kh * p.Crun;
  p.To = To; p.Ho = Ho; p.Wo = Wo; p.Ti = Tp; p.Hi = Hp; p.Wi = Wp;
  p.kt = kt; p.kh = kh; p.kw = 1;
  p.st = st; p.sh = sh; p.sw = sw; p.pt = 0; p.ph = 0; p.pw = 0; p.dt = 1; p.dh = 1; p.dw = 1;
  p.in_n_stride = (long long)Tp * Hp * Wp * Cp;
  p.pix = Cp;
  p.ldw = p.Ktot;
  p.ldo = Cout;
  p.alpha = 1.0f;
  p.epi = epilogue & ~TRACE_EPI_RESIDUAL;
  // the run must stay inside the padded row: checked above via (Wo-1)*sw+kwp <= Wp
  if (kt == ST_KT && kh == ST_KH && kwp == 8 && Cp == 4 && Cout == 64 && sh == 2 && sw == 2 && Ho % 16 == 0 &&
      Wo % 16 == 0 && !(epilogue & TRACE_EPI_OUT_F32) && !g_force_v2 && g_tune.tile == 0) {
    // TRACE's stem shape: LDS-resident input patch (stem16_bf16_kernel); anything else takes the generic loop
    TRACE_LDS_OPT_IN((&stem16_bf16_kernel<0>), ST_LDS, "igemm");
    const long long nwg = (long long)N * To * (Ho / 16) * (Wo / 16);
    TRACE_REQUIRE(nwg < 0x7fffffffLL, "conv3d_stem: grid too large");
    hipLaunchKernelGGL(stem16_bf16_kernel<0>, dim3((unsigned)nwg), dim3(256), ST_LDS, (hipStream_t)stream, p);
    return check_launch("conv3d_stem16");
  }
  return launch_igemm(p, 1, (hipStream_t)stream);
}

extern "C" int trace_conv3d_stem16_f32(const float* clip, int N, int C, int T, int H, int W, int in_ntchw,
                                       const uint16_t* w, const float* bias, int st, int To, int Ho, int Wo, int epilogue,
                                       void* out, trace_stream_t stream) {
  TRACE_REQUIRE(N >= 0 && C >= 1 && C <= 3 && T > 0 && H > 0 && W > 0 && st > 0, "conv3d_stem16_f32: bad shape");
  TRACE_REQUIRE(Ho > 0 && Wo > 0 && To > 0 && Ho % 16 == 0 && Wo % 16 == 0, "conv3d_stem16_f32: Ho, Wo must be multiples of 16");
  TRACE_REQUIRE(Ho == (H + 6 - 7) / 2 + 1 && Wo == (W + 6 - 7) / 2 + 1 && (To - 1) * st + 5 <= T + 4,
                "conv3d_stem16_f32: output (%d,%d,%d) does not match a 5x7x7 / pad (2,3,3) / stride (%d,2,2) conv of (%d,%d,%d)",
                To, Ho, Wo, st, T, H, W);
  TRACE_REQUIRE(!(epilogue & (TRACE_EPI_OUT_F32 | TRACE_EPI_RESIDUAL)), "conv3d_stem16_f32: bf16 output, no residual");
  long long M = (long long)N * To * Ho * Wo;
  TRACE_REQUIRE(M < 0x7fffffffLL, "conv3d_stem16_f32: too many output positions");
  if (M == 0) return 0;
  TRACE_REQUIRE(clip && w && out, "conv3d_stem16_f32: null pointer");
  TRACE_REQUIRE((((uintptr_t)w | (uintptr_t)out | (uintptr_t)bias) & 15) == 0 && ((uintptr_t)clip & 3) == 0,
                "conv3d_stem16_f32: pointers misaligned");
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  p.in_f32 = clip;
  p.f_C = C; p.f_T = T; p.f_H = H; p.f_W = W; p.f_ntchw = in_ntchw ? 1 : 0;
  p.w = w;
  p.bias = bias;
  p.out = out;
  p.M = (int)M;
  p.Cout = 64;
  p.Crun = 32;
  p.Ktot = ST_KT * ST_KH * 32;
  p.To = To; p.Ho = Ho; p.Wo = Wo;
  p.kt = ST_KT; p.kh = ST_KH; p.kw = 1;
  p.st = st; p.sh = 2; p.sw = 2;
  p.ldw = p.Ktot;
  p.ldo = 64;
  p.alpha = 1.0f;
  p.epi = epilogue;
  TRACE_LDS_OPT_IN((&stem16_bf16_kernel<1>), ST_LDS, "igemm");
  const long long nwg = (long long)N * To * (Ho / 16) * (Wo / 16);
  TRACE_REQUIRE(nwg < 0x7fffffffLL, "conv3d_stem16_f32: grid too large");
  hipLaunchKernelGGL(stem16_bf16_kernel<1>, dim3((unsigned)nwg), dim3(256), ST_LDS, (hipStream_t)stream, p);
  return check_launch("conv3d_stem16_f32");
}

extern "C" int trace_conv3d_stem16_u8(const uint8_t* frames, int N, int T, int H, int W, int C, const float* mean,
                                      const uint16_t* w, const float* bias, int st, int To, int Ho, int Wo, int epilogue,
                                      void* out, trace_stream_t stream) {
  TRACE_REQUIRE(N >= 0 && C >= 1 && C <= 3 && T > 0 && H > 0 && W > 0 && st > 0, "conv3d_stem16_u8: bad shape");
  TRACE_REQUIRE(Ho > 0 && Wo > 0 && To > 0 && Ho % 16 == 0 && Wo % 16 == 0, "conv3d_stem16_u8: Ho, Wo must be multiples of 16");
  TRACE_REQUIRE(Ho == (H + 6 - 7) / 2 + 1 && Wo == (W + 6 - 7) / 2 + 1 && (To - 1) * st + 5 <= T + 4,
                "conv3d_stem16_u8: output (%d,%d,%d) does not match a 5x7x7 / pad (2,3,3) / stride (%d,2,2) conv of (%d,%d,%d)",
                To, Ho, Wo, st, T, H, W);
  TRACE_REQUIRE(!(epilogue & (TRACE_EPI_OUT_F32 | TRACE_EPI_RESIDUAL)), "conv3d_stem16_u8: bf16 output, no residual");
  long long M = (long long)N * To * Ho * Wo;
  TRACE_REQUIRE(M < 0x7fffffffLL, "conv3d_stem16_u8: too many output positions");
  if (M == 0) return 0;
  TRACE_REQUIRE(frames && w && out, "conv3d_stem16_u8: null pointer");
  TRACE_REQUIRE((((uintptr_t)w | (uintptr_t)out | (uintptr_t)bias) & 15) == 0, "conv3d_stem16_u8: pointers misaligned");
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  p.in_u8 = frames;
  for (int c = 0; c < 3; ++c) p.u8_mean[c] = (mean && c < C) ? mean[c] : 0.f;   // `mean`: C floats on the HOST (PIXEL_MEANS), or null
  p.f_C = C; p.f_T = T; p.f_H = H; p.f_W = W;
  p.w = w;
  p.bias = bias;
  p.out = out;
  p.M = (int)M;
  p.Cout = 64;
  p.Crun = 32;
  p.Ktot = ST_KT * ST_KH * 32;
  p.To = To; p.Ho = Ho; p.Wo = Wo;
  p.kt = ST_KT; p.kh = ST_KH; p.kw = 1;
  p.st = st; p.sh = 2; p.sw = 2;
  p.ldw = p.Ktot;
  p.ldo = 64;
  p.alpha = 1.0f;
  p.epi = epilogue;
  TRACE_LDS_OPT_IN((&stem16_bf16_kernel<2>), ST_LDS, "igemm");
  const long long nwg = (long long)N * To * (Ho / 16) * (Wo / 16);
  TRACE_REQUIRE(nwg < 0x7fffffffLL, "conv3d_stem16_u8: grid too large");
  hipLaunchKernelGGL(stem16_bf16_kernel<2>, dim3((unsigned)nwg), dim3(256), ST_LDS, (hipStream_t)stream, p);
  return check_launch("conv3d_stem16_u8");
}

static int gemm_impl(const uint16_t* A, const uint16_t* Bm, int M, int N, int K, int lda, int ldb, int ldc, int batch,
                     long long strideA, long long strideB, long long strideC, float alpha, const float* bias,
                     long long strideBias, const void* residual, int epilogue, void* C, uint16_t* C2, int epilogue2,
                     void* workspace, size_t workspace_bytes, trace_stream_t stream, const uint16_t* A2 = nullptr,
                     int lda2 = 0, int K1 = 0, int dry = 0) {
  TRACE_REQUIRE(M >= 0 && N >= 0 && K > 0 && batch >= 0, "gemm: bad shape M=%d N=%d K=%d", M, N, K);
  if (M == 0 || N == 0 || batch == 0) return 0;
  TRACE_REQUIRE(K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0, "gemm: K, lda, ldb must be multiples of 8");
  TRACE_REQUIRE(N % 4 == 0 && ldc % 4 == 0, "gemm: N and ldc must be multiples of 4");
  TRACE_REQUIRE(strideA % 8 == 0 && strideB % 8 == 0 && strideC % 4 == 0 && strideBias % 4 == 0,
                "gemm: batch strides misaligned");
  TRACE_REQUIRE(lda >= (A2 ? K1 : K) && ldb >= K && ldc >= N, "gemm: leading dimensions too small");
  TRACE_REQUIRE(batch <= 65535, "gemm: batch too large");
  TRACE_REQUIRE(A && Bm && C, "gemm: null pointer");
  TRACE_REQUIRE(!(epilogue & (TRACE_EPI_RESIDUAL | TRACE_EPI_RES_POST)) || residual,
                "gemm: residual flag without pointer");
  TRACE_REQUIRE((epilogue2 & ~(TRACE_EPI_RELU | TRACE_EPI_LEAKY)) == 0, "gemm: epilogue2 takes RELU / LEAKY only");
  TRACE_REQUIRE((((uintptr_t)A | (uintptr_t)Bm | (uintptr_t)C | (uintptr_t)C2 | (uintptr_t)residual | (uintptr_t)bias) & 15) == 0,
                "gemm: pointers must be 16-byte aligned");
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  p.in = A;
  p.w = Bm;
  p.bias = bias;
  p.bs_bias = strideBias;
  p.res = (const uint16_t*)residual;
  p.out = C;
  p.out2 = C2;
  p.epi2 = epilogue2;
  p.M = M;
  p.Cout = N;
  p.Crun = K;
  p.Ktot = K;
  p.To = 1; p.Ho = 1; p.Wo = M; p.Ti = 1; p.Hi = 1; p.Wi = M;
  p.kt = 1; p.kh = 1; p.kw = 1;
  p.st = 1; p.sh = 1; p.sw = 1; p.dt = 1; p.dh = 1; p.dw = 1;
  p.in_n_stride = (long long)M * lda;  // single 'clip': n is always 0
  p.pix = lda;
  p.ldw = ldb;
  p.ldo = ldc;
  p.bs_in = strideA; p.bs_w = strideB; p.bs_out = strideC;
  p.alpha = alpha;
  p.epi = epilogue & ~TRACE_EPI_WTILED;
  p.w_tiled = (epilogue & TRACE_EPI_WTILED) ? 1 : 0;
  TRACE_REQUIRE(!p.w_tiled || (batch == 1 && ldb == K), "gemm: TRACE_EPI_WTILED takes one dense pre-tiled weight matrix");
  // split-K partials are dense [M][N]: only for the plain (non-batched) case
  p.ws = (batch == 1 && !((uintptr_t)workspace & 15)) ? workspace : nullptr;
  p.ws_bytes = workspace_bytes;
  if (A2) {
    TRACE_REQUIRE(batch == 1 && K1 > 0 && K1 < K && K1 % BK == 0 && (K - K1) % BK == 0 && lda2 >= K - K1 && lda2 % 8 == 0 &&
                      ((uintptr_t)A2 & 15) == 0 && !p.w_tiled && !g_force_v2 && g_tune.tile != 10,
                  "gemm: two-source form needs K1 and K - K1 multiples of 64, one batch, row-major weights");
    p.in2 = A2;
    p.pix2 = lda2;
    p.k_switch = K1 / BK;
  }
  p.dry = dry;
  return launch_igemm(p, batch, (hipStream_t)stream);
}

// Split-K scratch wanted by trace_gemm_bf16_nt / _ex for this problem (see trace_conv3d_workspace_bytes); 0 when the
// launch rules do not split K (batched problems never do).
extern "C" size_t trace_gemm_workspace_bytes(int M, int N, int K, int lda, int ldb, int ldc, int batch, int epilogue) {
  static const uint4 dummy = {};
  const uint16_t* a = reinterpret_cast<const uint16_t*>(&dummy);
  g_dry_bytes = 0;
  const int rc = gemm_impl(a, a, M, N, K, lda, ldb, ldc, batch, 0, 0, 0, 1.0f, nullptr, 0,
                           (epilogue & (TRACE_EPI_RESIDUAL | TRACE_EPI_RES_POST)) ? a : nullptr, epilogue, (void*)a, nullptr, 0,
                           (void*)a, (size_t)-1, nullptr, nullptr, 0, 0, 1);
  return rc ? 0 : g_dry_bytes;
}

extern "C" int trace_gemm_bf16_nt(const uint16_t* A, const uint16_t* Bm, int M, int N, int K, int lda,
                                  int ldb, int ldc, int batch, long long strideA, long long strideB,
                                  long long strideC, float alpha, const float* bias,
                                  const uint16_t* residual, int epilogue, void* C, void* workspace,
                                  size_t workspace_bytes, trace_stream_t stream) {
  TRACE_REQUIRE(!(epilogue & TRACE_EPI_RES_F32), "gemm: TRACE_EPI_RES_F32 needs trace_gemm_bf16_nt_ex");
  return gemm_impl(A, Bm, M, N, K, lda, ldb, ldc, batch, strideA, strideB, strideC, alpha, bias, 0, residual, epilogue,
                   C, nullptr, 0, workspace, workspace_bytes, stream);
}

extern "C" int trace_gemm_bf16_nt_ex(const uint16_t* A, const uint16_t* Bm, int M, int N, int K, int lda, int ldb,
                                     int ldc, int batch, long long strideA, long long strideB, long long strideC,
                                     float alpha, const float* bias, long long strideBias, const void* residual,
                                     int epilogue, void* C, uint16_t* C2, int epilogue2, void* workspace,
                                     size_t workspace_bytes, trace_stream_t stream) {
  return gemm_impl(A, Bm, M, N, K, lda, ldb, ldc, batch, strideA, strideB, strideC, alpha, bias, strideBias, residual,
                   epilogue, C, C2, epilogue2, workspace, workspace_bytes, stream);
}

extern "C" int trace_gemm_bf16_nt_dual(const uint16_t* A, int lda, int K1, const uint16_t* A2, int lda2, int K2,
                                       const uint16_t* Bm, int ldb, int M, int N, int ldc, const float* bias,
                                       const void* residual, int epilogue, void* C, trace_stream_t stream) {
  TRACE_REQUIRE(A2 && K1 > 0 && K2 > 0, "gemm_dual: bad shape");
  return gemm_impl(A, Bm, M, N, K1 + K2, lda, ldb, ldc, 1, 0, 0, 0, 1.0f, bias, 0, residual, epilogue, C, nullptr, 0, nullptr, 0,
                   stream, A2, lda2, K1);
}

extern "C" int trace_gemm_bf16_nt_dual_ex(const uint16_t* A, int lda, int K1, const uint16_t* A2, int lda2, int K2,
                                          const uint16_t* Bm, int ldb, int M, int N, int ldc, const float* bias,
                                          const void* residual, int epilogue, void* C, uint16_t* C2, int epilogue2,
                                          trace_stream_t stream) {
  TRACE_REQUIRE(A2 && K1 > 0 && K2 > 0, "gemm_dual: bad shape");
  return gemm_impl(A, Bm, M, N, K1 + K2, lda, ldb, ldc, 1, 0, 0, 0, 1.0f, bias, 0, residual, epilogue, C, C2, epilogue2, nullptr, 0,
                   stream, A2, lda2, K1);
}
