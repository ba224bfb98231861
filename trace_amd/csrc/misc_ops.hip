// Layout / pooling helpers of the I3D path on channels-last bf16 (gfx950).
//   clip_to_ndhwc   lib/modeling_rel/model_builder_rel.py:978 (permute) fused with
//                   the f32->bf16 cast and the stem's zero halo
//   maxpool3d       lib/i3d/resnet3d.py:732-737 (+ mmcv NonLocal3d sub_sample pool)
//   transpose       g^T for the non-local block's second matmul
#include "common.h"

namespace trace {

// One thread per output pixel (n,t,h,w) of the padded channels-last clip.
__global__ __launch_bounds__(256) void clip_to_ndhwc_kernel(
    const float* __restrict__ in, int N, int C, int T, int H, int W, int in_ntchw, int Cp, int pt0,
    int ph0, int pw0, int Tp, int Hp, int Wp, uint16_t* __restrict__ out) {
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long total = (long long)N * Tp * Hp * Wp;
  if (idx >= total) return;
  int wp = (int)(idx % Wp);
  long long r = idx / Wp;
  int hp = (int)(r % Hp);
  r /= Hp;
  int tp = (int)(r % Tp);
  int n = (int)(r / Tp);
  int t = tp - pt0, h = hp - ph0, w = wp - pw0;
  bool inside = (unsigned)t < (unsigned)T && (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
  uint16_t* o = out + idx * Cp;
  for (int c = 0; c < Cp; ++c) {
    float v = 0.f;
    if (inside && c < C) {
      long long src = in_ntchw ? ((((long long)n * T + t) * C + c) * H + h) * W + w
                               : ((((long long)n * C + c) * T + t) * H + h) * W + w;
      v = in[src];
    }
    o[c] = f32_to_bf16(v);
  }
}

// Cp == 4 fast path: a thread converts two adjacent pixels (2 x 8 bytes -> one 16-byte store); the three
// planes are read as coalesced float streams.
__global__ __launch_bounds__(256) void clip_to_ndhwc4_kernel(
    const float* __restrict__ in, int N, int C, int T, int H, int W, int in_ntchw, int pt0, int ph0, int pw0, int Tp,
    int Hp, int Wp, uint16_t* __restrict__ out) {
  const int Wh = Wp >> 1;
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long total = (long long)N * Tp * Hp * Wh;
  if (idx >= total) return;
  int wq = (int)(idx % Wh);
  long long r = idx / Wh;
  int hp = (int)(r % Hp);
  r /= Hp;
  int tp = (int)(r % Tp);
  int n = (int)(r / Tp);
  const int t = tp - pt0, h = hp - ph0, w0 = 2 * wq - pw0;
  const bool row_in = (unsigned)t < (unsigned)T && (unsigned)h < (unsigned)H;
  float v[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  if (row_in) {
    const long long plane = (long long)H * W;
    const long long cstride = in_ntchw ? plane : (long long)T * plane;
    const long long base = in_ntchw ? (((long long)n * T + t) * C) * plane + (long long)h * W
                                    : ((long long)n * C * T + t) * plane + (long long)h * W;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int w = w0 + q;
      if ((unsigned)w < (unsigned)W) {
        for (int c = 0; c < C; ++c) v[q][c] = in[base + c * cstride + w];
      }
    }
  }
  uint4 o;
  o.x = pack_bf16x2(v[0][0], v[0][1]);
  o.y = pack_bf16x2(v[0][2], v[0][3]);
  o.z = pack_bf16x2(v[1][0], v[1][1]);
  o.w = pack_bf16x2(v[1][2], v[1][3]);
  *reinterpret_cast<uint4*>(out + idx * 8) = o;
}

// One thread per (output position, 8-channel group); implicit -inf padding.
__global__ __launch_bounds__(256) void maxpool3d_kernel(const uint16_t* __restrict__ in, int N, int Ti,
                                                        int Hi, int Wi, int C, int kt, int kh, int kw,
                                                        int st, int sh, int sw, int pt, int ph, int pw,
                                                        int To, int Ho, int Wo,
                                                        uint16_t* __restrict__ out) {
  const int cg = C / 8;
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long total = (long long)N * To * Ho * Wo * cg;
  if (idx >= total) return;
  int g = (int)(idx % cg);
  long long r = idx / cg;
  int wo = (int)(r % Wo);
  r /= Wo;
  int ho = (int)(r % Ho);
  r /= Ho;
  int to = (int)(r % To);
  int n = (int)(r / To);
  float m[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) m[j] = -__builtin_inff();
  for (int a = 0; a < kt; ++a) {
    int ti = to * st - pt + a;
    if ((unsigned)ti >= (unsigned)Ti) continue;
    for (int b = 0; b < kh; ++b) {
      int hi = ho * sh - ph + b;
      if ((unsigned)hi >= (unsigned)Hi) continue;
      for (int c = 0; c < kw; ++c) {
        int wi = wo * sw - pw + c;
        if ((unsigned)wi >= (unsigned)Wi) continue;
        const uint4 v = *reinterpret_cast<const uint4*>(
            in + ((((long long)n * Ti + ti) * Hi + hi) * Wi + wi) * C + g * 8);
        const uint32_t* u = &v.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          m[2 * j] = fmaxf(m[2 * j], __uint_as_float(u[j] << 16));
          m[2 * j + 1] = fmaxf(m[2 * j + 1], __uint_as_float(u[j] & 0xffff0000u));
        }
      }
    }
  }
  uint4 o;
  o.x = pack_bf16x2(m[0], m[1]);
  o.y = pack_bf16x2(m[2], m[3]);
  o.z = pack_bf16x2(m[4], m[5]);
  o.w = pack_bf16x2(m[6], m[7]);
  *reinterpret_cast<uint4*>(out + idx * 8) = o;
}

// (batch, rows, cols) -> (batch, cols, rows), 64x64 tiles through LDS.
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const uint16_t* __restrict__ in, int rows,
                                                             int cols, uint16_t* __restrict__ out,
                                                             int ld_out, long long out_bs) {
  __shared__ uint16_t tile[64][66];
  const uint16_t* src = in + (long long)blockIdx.z * rows * cols;
  uint16_t* dst = out + (long long)blockIdx.z * out_bs;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < rows && c < cols) ? src[(long long)r * cols + c] : (uint16_t)0;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    int c = c0 + i, r = r0 + tx;
    if (c < cols && r < rows) dst[(long long)c * ld_out + r] = tile[tx][i];
  }
}


// Non-local block, sub-sampled (mmcv NonLocal3d, call site lib/i3d/resnet3d.py:289; cfg
// i3d_nl_dot_product_...py:15-19): the theta|phi|g 1x1x1 convs run as ONE GEMM whose output rows are
// [theta | phi | g] (3c channels).  This kernel max-pools (1,2,2) the phi and g thirds and writes them in
// the layouts the two products read: phi_p (N, nkp, c) row-major (B operand of theta.phi^T) and g^T
// (N, c, nkp) (B operand of pairwise.g), key positions nkv..nkp-1 zero-filled (nkp = nkv rounded up to 8).
// Block = 64 key positions x 64 channels; g goes through an LDS tile for the transposed 16-byte stores.
__global__ __launch_bounds__(256) void nl_pool_split_kernel(const uint16_t* __restrict__ tpg, int T, int H, int W,
                                                            int c, int nkv, int nkp,
                                                            uint16_t* __restrict__ phi_p,
                                                            uint16_t* __restrict__ gT,
                                                            uint16_t* __restrict__ phiT) {
  __shared__ __attribute__((aligned(16))) uint16_t tile[64][72];
  __shared__ __attribute__((aligned(16))) uint16_t tile2[64][72];   // phi^T (reassociated product only)
  const int n = blockIdx.z;
  const int p0 = blockIdx.x * 64, ch0 = blockIdx.y * 64;
  const int Ho = H >> 1, Wo = W >> 1, C3 = 3 * c;
  const uint16_t* src = tpg + (long long)n * T * H * W * C3;
  for (int item = threadIdx.x; item < 512; item += 256) {
    const int pl = item >> 3, g8 = item & 7;
    const int p = p0 + pl;
    float mphi[8], mg[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) mphi[j] = mg[j] = 0.f;
    if (p < nkv) {
      const int wo = p % Wo, r = p / Wo, ho = r % Ho, t = r / Ho;
#pragma unroll
      for (int j = 0; j < 8; ++j) mphi[j] = mg[j] = -__builtin_inff();
#pragma unroll
      for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          const uint16_t* px = src + (((long long)t * H + 2 * ho + dy) * W + 2 * wo + dx) * C3 + ch0 + g8 * 8;
          const uint4 a = *reinterpret_cast<const uint4*>(px + c);
          const uint4 b = *reinterpret_cast<const uint4*>(px + 2 * c);
          const uint32_t* ua = &a.x;
          const uint32_t* ub = &b.x;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            mphi[2 * j] = fmaxf(mphi[2 * j], __uint_as_float(ua[j] << 16));
            mphi[2 * j + 1] = fmaxf(mphi[2 * j + 1], __uint_as_float(ua[j] & 0xffff0000u));
            mg[2 * j] = fmaxf(mg[2 * j], __uint_as_float(ub[j] << 16));
            mg[2 * j + 1] = fmaxf(mg[2 * j + 1], __uint_as_float(ub[j] & 0xffff0000u));
          }
        }
    }
    if (phiT) {
#pragma unroll
      for (int j = 0; j < 8; ++j) tile2[g8 * 8 + j][pl] = f32_to_bf16(mphi[j]);
    }
    if (phi_p && p < nkp) {
      uint4 o;
      o.x = pack_bf16x2(mphi[0], mphi[1]);
      o.y = pack_bf16x2(mphi[2], mphi[3]);
      o.z = pack_bf16x2(mphi[4], mphi[5]);
      o.w = pack_bf16x2(mphi[6], mphi[7]);
      *reinterpret_cast<uint4*>(phi_p + ((long long)n * nkp + p) * c + ch0 + g8 * 8) = o;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) tile[g8 * 8 + j][pl] = f32_to_bf16(mg[j]);
  }
  __syncthreads();
  for (int item = threadIdx.x; item < 512; item += 256) {
    const int ch = item >> 3, pg = item & 7;
    const int p = p0 + pg * 8;
    if (p < nkp) {  // nkp % 8 == 0: a group of 8 key positions is inside or outside as a whole
      *reinterpret_cast<uint4*>(gT + ((long long)n * c + ch0 + ch) * nkp + p) =
          *reinterpret_cast<const uint4*>(&tile[ch][pg * 8]);
      if (phiT)
        *reinterpret_cast<uint4*>(phiT + ((long long)n * c + ch0 + ch) * nkp + p) =
            *reinterpret_cast<const uint4*>(&tile2[ch][pg * 8]);
    }
  }
}

}  // namespace trace

using namespace trace;

extern "C" int trace_clip_to_ndhwc_bf16(const float* in, int N, int C, int T, int H, int W,
                                        int in_ntchw, int Cp, int pt0, int pt1, int ph0, int ph1,
                                        int pw0, int pw1, uint16_t* out, trace_stream_t stream) {
  TRACE_REQUIRE(N >= 0 && C > 0 && T > 0 && H > 0 && W > 0 && Cp >= C, "clip_to_ndhwc: bad shape");
  TRACE_REQUIRE(pt0 >= 0 && pt1 >= 0 && ph0 >= 0 && ph1 >= 0 && pw0 >= 0 && pw1 >= 0,
                "clip_to_ndhwc: negative padding");
  if (N == 0) return 0;
  TRACE_REQUIRE(in && out, "clip_to_ndhwc: null pointer");
  int Tp = T + pt0 + pt1, Hp = H + ph0 + ph1, Wp = W + pw0 + pw1;
  long long total = (long long)N * Tp * Hp * Wp;
  TRACE_REQUIRE(total < (1LL << 31) * 256, "clip_to_ndhwc: too large");
  if (Cp == 4 && C <= 4 && (Wp & 1) == 0 && (((uintptr_t)out) & 15) == 0) {
    const long long half = total / 2;
    hipLaunchKernelGGL(clip_to_ndhwc4_kernel, dim3((unsigned)((half + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       in, N, C, T, H, W, in_ntchw, pt0, ph0, pw0, Tp, Hp, Wp, out);
    return check_launch("clip_to_ndhwc_bf16");
  }
  hipLaunchKernelGGL(clip_to_ndhwc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, in, N, C, T, H, W, in_ntchw, Cp, pt0, ph0, pw0, Tp, Hp, Wp,
                     out);
  return check_launch("clip_to_ndhwc_bf16");
}

extern "C" int trace_maxpool3d_ndhwc_bf16(const uint16_t* in, int N, int Ti, int Hi, int Wi, int C,
                                          int kt, int kh, int kw, int st, int sh, int sw, int pt,
                                          int ph, int pw, uint16_t* out, trace_stream_t stream) {
  TRACE_REQUIRE(N >= 0 && Ti > 0 && Hi > 0 && Wi > 0 && C > 0 && C % 8 == 0,
                "maxpool3d: bad shape (C must be a multiple of 8)");
  TRACE_REQUIRE(kt > 0 && kh > 0 && kw > 0 && st > 0 && sh > 0 && sw > 0, "maxpool3d: bad kernel/stride");
  TRACE_REQUIRE(2 * pt <= kt && 2 * ph <= kh && 2 * pw <= kw, "maxpool3d: pad must be <= kernel/2");
  int To = (Ti + 2 * pt - kt) / st + 1, Ho = (Hi + 2 * ph - kh) / sh + 1, Wo = (Wi + 2 * pw - kw) / sw + 1;
  TRACE_REQUIRE(To > 0 && Ho > 0 && Wo > 0, "maxpool3d: empty output");
  if (N == 0) return 0;
  TRACE_REQUIRE(in && out, "maxpool3d: null pointer");
  long long total = (long long)N * To * Ho * Wo * (C / 8);
  hipLaunchKernelGGL(maxpool3d_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, in, N, Ti, Hi, Wi, C, kt, kh, kw, st, sh, sw, pt, ph, pw, To,
                     Ho, Wo, out);
  return check_launch("maxpool3d_ndhwc_bf16");
}

extern "C" int trace_transpose_bf16(const uint16_t* in, int batch, int rows, int cols, uint16_t* out,
                                    int ld_out, long long out_batch_stride, trace_stream_t stream) {
  TRACE_REQUIRE(batch >= 0 && rows >= 0 && cols >= 0, "transpose: bad shape");
  TRACE_REQUIRE(ld_out >= rows && out_batch_stride >= (long long)cols * ld_out, "transpose: bad output strides");
  if (batch == 0 || rows == 0 || cols == 0) return 0;
  TRACE_REQUIRE(batch <= 65535 && ceil_div(rows, 64) <= 65535, "transpose: grid too large");
  TRACE_REQUIRE(in && out, "transpose: null pointer");
  hipLaunchKernelGGL(transpose_bf16_kernel, dim3(ceil_div(cols, 64), ceil_div(rows, 64), batch),
                     dim3(256), 0, (hipStream_t)stream, in, rows, cols, out, ld_out, out_batch_stride);
  return check_launch("transpose_bf16");
}

extern "C" int trace_nl_pool_split2_bf16(const uint16_t* tpg, int N, int T, int H, int W, int c, uint16_t* phi_p,
                                         uint16_t* phiT, uint16_t* gT, int nkp, trace_stream_t stream) {
  TRACE_REQUIRE(N >= 0 && T > 0 && H >= 2 && W >= 2 && c > 0, "nl_pool_split: bad shape");
  TRACE_REQUIRE(c % 64 == 0, "nl_pool_split: inter channels %d must be a multiple of 64", c);
  const int nkv = T * (H / 2) * (W / 2);
  TRACE_REQUIRE(nkp >= nkv && nkp % 8 == 0 && nkp < nkv + 8, "nl_pool_split: nkp must be nkv rounded up to 8");
  if (N == 0) return 0;
  TRACE_REQUIRE(N <= 65535, "nl_pool_split: batch too large");
  TRACE_REQUIRE(tpg && (phi_p || phiT) && gT, "nl_pool_split: null pointer");
  TRACE_REQUIRE((((uintptr_t)tpg | (uintptr_t)phi_p | (uintptr_t)phiT | (uintptr_t)gT) & 15) == 0,
                "nl_pool_split: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(nl_pool_split_kernel, dim3(ceil_div(nkp, 64), c / 64, N), dim3(256), 0, (hipStream_t)stream, tpg,
                     T, H, W, c, nkv, nkp, phi_p, gT, phiT);
  return check_launch("nl_pool_split_bf16");
}

extern "C" int trace_nl_pool_split_bf16(const uint16_t* tpg, int N, int T, int H, int W, int c, uint16_t* phi_p,
                                        uint16_t* gT, int nkp, trace_stream_t stream) {
  TRACE_REQUIRE(N == 0 || phi_p, "nl_pool_split: null pointer");
  return trace_nl_pool_split2_bf16(tpg, N, T, H, W, c, phi_p, nullptr, gT, nkp, stream);
}
