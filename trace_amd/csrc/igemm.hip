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
//     into registers (zero-filled at the padding halo) and stores them to LDS;
//   * weights are pre-packed (Cout, taps*C) bf16 with the BN scale folded in, the
//     BN shift is the epilogue bias; residual add + ReLU are fused in the epilogue;
//   * D = W_tile (A operand, rows = Cout) x X_tile^T (B operand, cols = output
//     positions) on v_mfma_f32_16x16x32_bf16: a lane then owns 4 consecutive
//     output channels of one position -> 8-byte channels-last stores;
//   * 4 waves / workgroup (2 x 2), wave tile (WM*16 couts) x (WN*16 positions),
//     BK = 64, LDS double-buffered, XOR-swizzled 16-byte chunks (conflict-free
//     ds_read_b128), next tile's global loads issued before the MFMA block
//     (register-staged split pipeline), one barrier per K-step;
//   * the same kernel runs plain/batched GEMMs (1 tap, explicit row strides).
#include "common.h"

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
  size_t ws_bytes;
};

constexpr int BK = 64;

// byte offset of 16-byte chunk `chunk` (0..7) of row `r` in a [rows][64] bf16 tile
__device__ __forceinline__ int lds_off(int r, int chunk) {
  return r * 128 + ((chunk ^ (r & 7)) << 4);
}

template <int WM, int WN>
__global__ __launch_bounds__(256) void igemm_bf16_kernel(const IgemmParams p) {
  constexpr int TC = 32 * WM;  // cout rows per block
  constexpr int TP = 32 * WN;  // positions per block
  constexpr int A_BYTES = TC * 128;
  constexpr int B_BYTES = TP * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // [buf0: A | B][buf1: A | B]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware tile order: blocks b and b+8 share an XCD's L2; give each XCD a
  // contiguous run of tiles so neighbouring position tiles reuse the same weights.
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x;
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
  }
  // default: cout tile fastest (a position tile's inputs stay hot in L2); weight-streaming shapes
  // (few position tiles, huge K) flip it so the tiles sharing a weight panel run side by side.
  const int tile_n = p.m_fast ? bid / p.tiles_m : bid % p.tiles_n;
  const int tile_m = p.m_fast ? bid % p.tiles_m : bid / p.tiles_n;
  const int cout0 = tile_n * TC;
  const int pos0 = tile_m * TP;

  const uint16_t* gin = p.in + (long long)blockIdx.z * p.bs_in;
  const uint16_t* gw = p.w + (long long)blockIdx.z * p.bs_w;

  // ---- per-thread load assignment: 16-byte piece `kseg` of rows (tid>>3)+32*i
  const int kseg = tid & 7;
  const int lrow = tid >> 3;
  long long x_nbase[WN];
  int x_t0[WN], x_h0[WN], x_w0[WN];
#pragma unroll
  for (int i = 0; i < WN; ++i) {
    int m = pos0 + lrow + 32 * i;
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
      x_t0[i] = -0x40000000;  // never in bounds
      x_h0[i] = 0;
      x_w0[i] = 0;
    }
  }
  // K range of this block (split-K slices along blockIdx.y)
  const int ksteps_all = (p.Ktot + BK - 1) / BK;
  const int ks_begin = blockIdx.y * p.ks_per;
  const int ksteps = min(ksteps_all - ks_begin, p.ks_per);
  // running tap state of this thread's K piece
  int klin = ks_begin * BK + kseg * 8;
  int kc, ta, tb, tcw;
  {
    const int tap = klin / p.Crun;
    kc = klin - tap * p.Crun;  // channel offset inside the current tap run
    tcw = tap % p.kw;
    const int t2 = tap / p.kw;
    tb = t2 % p.kh;
    ta = t2 / p.kh;
  }

  uint4 xr[WN], wr[WM];
  auto load_tiles = [&]() {
    const bool kvalid = ta < p.kt;
#pragma unroll
    for (int i = 0; i < WN; ++i) {
      int ti = x_t0[i] + ta * p.dt, hi = x_h0[i] + tb * p.dh, wi = x_w0[i] + tcw * p.dw;
      bool ok = kvalid && (unsigned)ti < (unsigned)p.Ti && (unsigned)hi < (unsigned)p.Hi &&
                (unsigned)wi < (unsigned)p.Wi;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (ok) {
        long long off = x_nbase[i] + ((long long)(ti * p.Hi + hi) * p.Wi + wi) * p.pix + kc;
        v = *reinterpret_cast<const uint4*>(gin + off);
      }
      xr[i] = v;
    }
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      int row = cout0 + lrow + 32 * i;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (kvalid && row < p.Cout) v = *reinterpret_cast<const uint4*>(gw + (long long)row * p.ldw + klin);
      wr[i] = v;
    }
  };
  auto advance_k = [&]() {
    klin += BK;
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
  };
  auto store_tiles = [&](int buf) {
    char* a = smem + buf * (A_BYTES + B_BYTES);
    char* b = a + A_BYTES;
#pragma unroll
    for (int i = 0; i < WM; ++i) *reinterpret_cast<uint4*>(a + lds_off(lrow + 32 * i, kseg)) = wr[i];
#pragma unroll
    for (int i = 0; i < WN; ++i) *reinterpret_cast<uint4*>(b + lds_off(lrow + 32 * i, kseg)) = xr[i];
  };

  f32x4 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  load_tiles();
  store_tiles(0);
  __syncthreads();

  const int fr = lane & 15, fq = lane >> 4;
  for (int ks = 0; ks < ksteps; ++ks) {
    const int cur = ks & 1;
    if (ks + 1 < ksteps) {
      advance_k();
      load_tiles();  // in flight during the MFMA block below
    }
    const char* a = smem + cur * (A_BYTES + B_BYTES);
    const char* b = a + A_BYTES;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 af[WM], bfr[WN];
#pragma unroll
      for (int i = 0; i < WM; ++i)
        af[i] = *reinterpret_cast<const bf16x8*>(a + lds_off(wm * (16 * WM) + i * 16 + fr, s * 4 + fq));
#pragma unroll
      for (int j = 0; j < WN; ++j)
        bfr[j] = *reinterpret_cast<const bf16x8*>(b + lds_off(wn * (16 * WN) + j * 16 + fr, s * 4 + fq));
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    if (ks + 1 < ksteps) {
      store_tiles(cur ^ 1);
      __syncthreads();
    }
  }

  // ---- split-K: raw partial sums, the epilogue runs in igemm_splitk_reduce_kernel
  if (p.ksplit > 1) {
    float* slab = p.partial + (long long)blockIdx.y * p.M * p.Cout;
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int m = pos0 + wn * (16 * WN) + j * 16 + fr;
      if (m >= p.M) continue;
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        const int c = cout0 + wm * (16 * WM) + i * 16 + fq * 4;
        if (c >= p.Cout) continue;
        *reinterpret_cast<float4*>(slab + (long long)m * p.Cout + c) =
            make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      }
    }
    return;
  }
  // ---- epilogue: D[row = cout (fq*4+r)][col = position fr]
  const bool relu = p.epi & TRACE_EPI_RELU;
  const bool leaky = p.epi & TRACE_EPI_LEAKY;
  const bool res_post = (p.epi & TRACE_EPI_RES_POST) && p.res;
  const bool has_res = (p.epi & TRACE_EPI_RESIDUAL) && p.res;
  const bool out_f32 = p.epi & TRACE_EPI_OUT_F32;
  const long long obase = (long long)blockIdx.z * p.bs_out;
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    const int m = pos0 + wn * (16 * WN) + j * 16 + fr;
    if (m >= p.M) continue;
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      const int c = cout0 + wm * (16 * WM) + i * 16 + fq * 4;
      if (c >= p.Cout) continue;  // Cout % 4 == 0 is required by the host wrapper
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] * p.alpha;
      if (p.bias) {
        const float4 bv = *reinterpret_cast<const float4*>(p.bias + c);
        v[0] += bv.x;
        v[1] += bv.y;
        v[2] += bv.z;
        v[3] += bv.w;
      }
      const long long o = obase + (long long)m * p.ldo + c;
      if (has_res) {
        const uint2 rv = *reinterpret_cast<const uint2*>(p.res + o);
        v[0] += __uint_as_float(rv.x << 16);
        v[1] += __uint_as_float(rv.x & 0xffff0000u);
        v[2] += __uint_as_float(rv.y << 16);
        v[3] += __uint_as_float(rv.y & 0xffff0000u);
      }
      if (relu) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
      }
      if (leaky) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.1f * v[r];
      }
      if (res_post) {
        const uint2 rv = *reinterpret_cast<const uint2*>(p.res + o);
        v[0] += __uint_as_float(rv.x << 16);
        v[1] += __uint_as_float(rv.x & 0xffff0000u);
        v[2] += __uint_as_float(rv.y << 16);
        v[3] += __uint_as_float(rv.y & 0xffff0000u);
      }
      if (out_f32) {
        *reinterpret_cast<float4*>((float*)p.out + o) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
        uint2 ov;
        ov.x = pack_bf16x2(v[0], v[1]);
        ov.y = pack_bf16x2(v[2], v[3]);
        *reinterpret_cast<uint2*>((uint16_t*)p.out + o) = ov;
      }
    }
  }
}

// Sum the split-K slabs in slice order (deterministic) and apply the epilogue.
__global__ __launch_bounds__(256) void igemm_splitk_reduce_kernel(const IgemmParams p) {
  const int cq = p.Cout >> 2;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)p.M * cq) return;
  const int m = (int)(idx / cq), c = (int)(idx % cq) * 4;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  for (int z = 0; z < p.ksplit; ++z) {
    const float4 t = *reinterpret_cast<const float4*>(p.partial + ((long long)z * p.M + m) * p.Cout + c);
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
    const uint2 rv = *reinterpret_cast<const uint2*>(p.res + o);
    v[0] += __uint_as_float(rv.x << 16);
    v[1] += __uint_as_float(rv.x & 0xffff0000u);
    v[2] += __uint_as_float(rv.y << 16);
    v[3] += __uint_as_float(rv.y & 0xffff0000u);
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
}

template <int WM, int WN>
static int launch_cfg(IgemmParams& p, int batch, hipStream_t s) {
  constexpr int TC = 32 * WM, TP = 32 * WN;
  p.tiles_n = ceil_div(p.Cout, TC);
  p.tiles_m = ceil_div(p.M, TP);
  long long nblk = (long long)p.tiles_m * p.tiles_n;
  if (nblk > 0x7fffffffLL) {
    set_error("igemm: grid too large");
    return 1;
  }
  size_t lds = 2 * (size_t)(TC + TP) * 128;
  const int ksteps = ceil_div(p.Ktot, BK);
  p.m_fast = (p.tiles_m <= 4 && ksteps >= 64) ? 1 : 0;
  // split-K when the tile grid cannot fill 256 CUs and K is deep (weight-streaming FC layers,
  // single-clip convs): slices write f32 slabs to the caller's workspace, a second launch reduces.
  p.ksplit = 1;
  p.ks_per = ksteps;
  if (batch == 1 && p.ws && nblk < 256 && ksteps >= 8) {
    long long want = (768 + nblk - 1) / nblk;
    long long cap_ws = (long long)(p.ws_bytes / ((size_t)p.M * p.Cout * sizeof(float)));
    long long ks = want;
    if (ks > ksteps / 4) ks = ksteps / 4;
    if (ks > cap_ws) ks = cap_ws;
    if (ks > 1) {
      p.ks_per = ceil_div(ksteps, ks);
      p.ksplit = ceil_div(ksteps, p.ks_per);
      p.partial = (float*)p.ws;
    }
  }
  hipLaunchKernelGGL((igemm_bf16_kernel<WM, WN>), dim3((unsigned)nblk, p.ksplit, batch), dim3(256), lds, s, p);
  if (p.ksplit > 1) {
    long long n4 = (long long)p.M * (p.Cout >> 2);
    hipLaunchKernelGGL(igemm_splitk_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, p);
  }
  return check_launch("igemm_bf16");
}

static int launch_igemm(IgemmParams& p, int batch, hipStream_t s) {
  if (p.M <= 0 || p.Cout <= 0 || batch <= 0) return 0;
  // Tile choice: keep >= ~2 workgroups per CU in flight when the problem allows.
  auto blocks = [&](int tc, int tp) { return (long long)ceil_div(p.Cout, tc) * ceil_div(p.M, tp) * batch; };
  if (p.Cout <= 64) return launch_cfg<2, 4>(p, batch, s);        // 64 couts x 128 positions
  if (p.ws && batch == 1 && p.M <= 256 && p.Ktot >= 8192) return launch_cfg<4, 2>(p, batch, s);  // weight streaming + split-K
  if (blocks(128, 128) >= 512) return launch_cfg<4, 4>(p, batch, s);
  if (blocks(128, 64) >= 384) return launch_cfg<4, 2>(p, batch, s);
  return launch_cfg<2, 2>(p, batch, s);                           // 64 x 64: small problems
}

}  // namespace trace

using namespace trace;

extern "C" int trace_conv3d_igemm_bf16(const uint16_t* in, int N, int Ti, int Hi, int Wi, int Cin,
                                       const uint16_t* w, const float* bias, int Cout, int kt,
                                       int kh, int kw, int st, int sh, int sw, int pt, int ph,
                                       int pw, int dt, int dh, int dw, const uint16_t* residual,
                                       int epilogue, void* out, void* workspace,
                                       size_t workspace_bytes, trace_stream_t stream) {
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
  p.epi = epilogue;
  p.ws = ((uintptr_t)workspace & 15) ? nullptr : workspace;
  p.ws_bytes = workspace_bytes;
  return launch_igemm(p, 1, (hipStream_t)stream);
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
  return launch_igemm(p, 1, (hipStream_t)stream);
}

extern "C" int trace_gemm_bf16_nt(const uint16_t* A, const uint16_t* Bm, int M, int N, int K, int lda,
                                  int ldb, int ldc, int batch, long long strideA, long long strideB,
                                  long long strideC, float alpha, const float* bias,
                                  const uint16_t* residual, int epilogue, void* C, void* workspace,
                                  size_t workspace_bytes, trace_stream_t stream) {
  TRACE_REQUIRE(M >= 0 && N >= 0 && K > 0 && batch >= 0, "gemm: bad shape M=%d N=%d K=%d", M, N, K);
  if (M == 0 || N == 0 || batch == 0) return 0;
  TRACE_REQUIRE(K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0, "gemm: K, lda, ldb must be multiples of 8");
  TRACE_REQUIRE(N % 4 == 0 && ldc % 4 == 0, "gemm: N and ldc must be multiples of 4");
  TRACE_REQUIRE(strideA % 8 == 0 && strideB % 8 == 0 && strideC % 4 == 0, "gemm: batch strides misaligned");
  TRACE_REQUIRE(lda >= K && ldb >= K && ldc >= N, "gemm: leading dimensions too small");
  TRACE_REQUIRE(batch <= 65535, "gemm: batch too large");
  TRACE_REQUIRE(A && Bm && C, "gemm: null pointer");
  TRACE_REQUIRE(!(epilogue & (TRACE_EPI_RESIDUAL | TRACE_EPI_RES_POST)) || residual,
                "gemm: residual flag without pointer");
  TRACE_REQUIRE((((uintptr_t)A | (uintptr_t)Bm | (uintptr_t)C | (uintptr_t)residual | (uintptr_t)bias) & 15) == 0,
                "gemm: pointers must be 16-byte aligned");
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  p.in = A;
  p.w = Bm;
  p.bias = bias;
  p.res = residual;
  p.out = C;
  p.M = M;
  p.Cout = N;
  p.Crun = K;
  p.Ktot = K;
  p.To = 1; p.Ho = 1; p.Wo = M; p.Ti = 1; p.Hi = 1; p.Wi = M;
  p.kt = 1; p.kh = 1; p.kw = 1;
  p.st = 1; p.sh = 1; p.sw = 1; p.dt = 1; p.dh = 1; p.dw = 1;
  p.in_n_stride = 0;
  p.pix = lda;
  p.ldw = ldb;
  p.ldo = ldc;
  p.bs_in = strideA; p.bs_w = strideB; p.bs_out = strideC;
  p.alpha = alpha;
  p.epi = epilogue;
  // split-K partials are dense [M][N]: only for the plain (non-batched) case
  p.ws = (batch == 1 && !((uintptr_t)workspace & 15)) ? workspace : nullptr;
  p.ws_bytes = workspace_bytes;
  return launch_igemm(p, batch, (hipStream_t)stream);
}
