// Context-aggregation kernels of the TRACE relation head (gfx950).
//
//   temporal attention   lib/combine_tree/cluster_tree3d.py:337-366
//   fused pair attention lib/modeling_rel/model_builder_rel.py:1524-1612
//   tree-GRU             lib/combine_tree/cluster_tree.py:163-204, 281-330
#include "common.h"
#include <stdlib.h>

namespace trace {

// ---------------------------------------------------------------------------
// Temporal attention: one workgroup per object/region.
// ---------------------------------------------------------------------------
constexpr int TA_MAX_T = 128;

__global__ __launch_bounds__(256) void temporal_attention_kernel(
    const float* __restrict__ qs, const float* __restrict__ qo, const float* __restrict__ o, int T,
    int D, int K, float inv_sqrt_d, float* __restrict__ out, uint16_t* __restrict__ out_bf16) {
  __shared__ float sc[TA_MAX_T];
  const int n = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* q = qs + (long long)n * D;
  for (int t = wave; t < T; t += 4) {
    const float* k = qo + ((long long)n * T + t) * D;
    float s = 0.f;
    for (int d = lane; d < D; d += 64) s += q[d] * k[d];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) sc[t] = s * inv_sqrt_d;  // cluster_tree3d.py:348
  }
  __syncthreads();
  float mx = -__builtin_inff();
  for (int t = 0; t < T; ++t) mx = fmaxf(mx, sc[t]);
  float sum = 0.f;
  for (int t = 0; t < T; ++t) sum += expf(sc[t] - mx);
  const float inv = 1.0f / sum;
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    float acc = 0.f;
    for (int t = 0; t < T; ++t) acc += (expf(sc[t] - mx) * inv) * o[((long long)n * T + t) * K + k];
    if (out) out[(long long)n * K + k] = acc;  // cluster_tree3d.py:349-350
    if (out_bf16) out_bf16[(long long)n * K + k] = f32_to_bf16(acc);
  }
}

// ---------------------------------------------------------------------------
// Fused pair attention.  One workgroup per (subject, object) pair:
//   union box -> adaptive ROIAlign of the K-map and V-map (channels-last bf16,
//   L2-resident) -> q_s.K / q_o.K per cell -> softmax over the cells -> s_map,
//   o_map = V * softmax, v_map = V.  The pooled K-map never reaches HBM.
// A lane owns 8 consecutive channels (16-byte loads); D/8 lanes share a cell.
// ---------------------------------------------------------------------------
struct PGeom {
  int batch;
  float start_w, start_h, bin_w, bin_h;
  int grid_h, grid_w;
};

constexpr int PA_MAX_CELLS = 64;

__global__ __launch_bounds__(256) void pair_attention_kernel(
    const uint16_t* __restrict__ kmap, const uint16_t* __restrict__ vmap, int B, int H, int W, int D,
    const float* __restrict__ rel_rois, const int32_t* __restrict__ sbj, const int32_t* __restrict__ obj,
    const float* __restrict__ q_s, const float* __restrict__ q_o, float scale, int PS, int so_stride,
    uint16_t* __restrict__ s_map, uint16_t* __restrict__ o_map, uint16_t* __restrict__ v_map) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sc_s = (float*)smem;              // [PA_MAX_CELLS]
  float* sc_o = sc_s + PA_MAX_CELLS;       // [PA_MAX_CELLS]
  uint16_t* vbuf = (uint16_t*)(sc_o + PA_MAX_CELLS);  // [cells][D]

  const int p = blockIdx.x;
  const int cells = PS * PS;
  const int lpb = D >> 3;           // lanes per cell (power of two, <= 64)
  const int bpr = 256 / lpb;        // cells per round
  const int cg = threadIdx.x % lpb; // channel group of this lane
  const int slot = threadIdx.x / lpb;

  // model_builder_rel.py:1588-1594 union roi; ROIAlign geometry as roi_ops.hip
  const float* r = rel_rois + (long long)p * 5;
  PGeom g;
  g.batch = min(max((int)r[0], 0), B - 1);
  g.start_w = r[1] * scale;
  g.start_h = r[2] * scale;
  {
    float end_w = r[3] * scale, end_h = r[4] * scale;
    float roi_w = fmaxf(end_w - g.start_w, 1.0f), roi_h = fmaxf(end_h - g.start_h, 1.0f);
    g.bin_h = roi_h / (float)PS;
    g.bin_w = roi_w / (float)PS;
    g.grid_h = (int)ceilf(roi_h / (float)PS);  // sampling_ratio = 0 (model_builder_rel.py:1343)
    g.grid_w = (int)ceilf(roi_w / (float)PS);
  }
  const float inv_count = 1.0f / (float)(g.grid_h * g.grid_w);
  const uint16_t* kimg = kmap + (long long)g.batch * H * W * D + cg * 8;
  const uint16_t* vimg = vmap + (long long)g.batch * H * W * D + cg * 8;

  float qs[8], qo[8];
  {
    const float* a = q_s + (long long)sbj[p] * D + cg * 8;
    const float* b = q_o + (long long)obj[p] * D + cg * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      qs[j] = a[j];
      qo[j] = b[j];
    }
  }

  for (int cell0 = 0; cell0 < cells; cell0 += bpr) {
    const int cell = cell0 + slot;
    const bool live = cell < cells;
    float ka[8], va[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) ka[j] = va[j] = 0.f;
    if (live) {
      const int ph = cell / PS, pw = cell - ph * PS;
      for (int iy = 0; iy < g.grid_h; ++iy) {
        float y = g.start_h + (float)ph * g.bin_h + ((float)iy + .5f) * g.bin_h / (float)g.grid_h;
        for (int ix = 0; ix < g.grid_w; ++ix) {
          float x = g.start_w + (float)pw * g.bin_w + ((float)ix + .5f) * g.bin_w / (float)g.grid_w;
          if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) continue;
          if (y <= 0.f) y = 0.f;
          if (x <= 0.f) x = 0.f;
          int y_low = (int)y, x_low = (int)x, y_high, x_high;
          float yy = y, xx = x;
          if (y_low >= H - 1) { y_high = y_low = H - 1; yy = (float)y_low; } else { y_high = y_low + 1; }
          if (x_low >= W - 1) { x_high = x_low = W - 1; xx = (float)x_low; } else { x_high = x_low + 1; }
          float ly = yy - (float)y_low, lx = xx - (float)x_low, hy = 1.f - ly, hx = 1.f - lx;
          float w4[4] = {hy * hx, hy * lx, ly * hx, ly * lx};
          int pos[4] = {y_low * W + x_low, y_low * W + x_high, y_high * W + x_low, y_high * W + x_high};
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const uint4 kv = *reinterpret_cast<const uint4*>(kimg + (long long)pos[c] * D);
            const uint4 vv = *reinterpret_cast<const uint4*>(vimg + (long long)pos[c] * D);
            const uint32_t* ku = &kv.x;
            const uint32_t* vu = &vv.x;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              ka[2 * j] += w4[c] * __uint_as_float(ku[j] << 16);
              ka[2 * j + 1] += w4[c] * __uint_as_float(ku[j] & 0xffff0000u);
              va[2 * j] += w4[c] * __uint_as_float(vu[j] << 16);
              va[2 * j + 1] += w4[c] * __uint_as_float(vu[j] & 0xffff0000u);
            }
          }
        }
      }
    }
    float ds = 0.f, dob = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      ka[j] *= inv_count;
      va[j] *= inv_count;
      ds += qs[j] * ka[j];   // model_builder_rel.py:1606
      dob += qo[j] * ka[j];
    }
    for (int off = lpb >> 1; off > 0; off >>= 1) {
      ds += __shfl_xor(ds, off);
      dob += __shfl_xor(dob, off);
    }
    if (live) {
      if (cg == 0) {
        sc_s[cell] = ds;
        sc_o[cell] = dob;
      }
      uint4 pk;
      pk.x = pack_bf16x2(va[0], va[1]);
      pk.y = pack_bf16x2(va[2], va[3]);
      pk.z = pack_bf16x2(va[4], va[5]);
      pk.w = pack_bf16x2(va[6], va[7]);
      *reinterpret_cast<uint4*>(vbuf + (long long)cell * D + cg * 8) = pk;
    }
  }
  __syncthreads();
  // softmax over the cells (model_builder_rel.py:1607-1609)
  float ms = -__builtin_inff(), mo = -__builtin_inff();
  for (int c = 0; c < cells; ++c) {
    ms = fmaxf(ms, sc_s[c]);
    mo = fmaxf(mo, sc_o[c]);
  }
  float ss = 0.f, so = 0.f;
  for (int c = 0; c < cells; ++c) {
    ss += expf(sc_s[c] - ms);
    so += expf(sc_o[c] - mo);
  }
  const float is = 1.0f / ss, io = 1.0f / so;
  const long long obase = (long long)p * cells * D;
  for (int idx = threadIdx.x; idx < cells * lpb; idx += blockDim.x) {
    const int cell = idx / lpb, c8 = idx - cell * lpb;
    const uint4 pk = *reinterpret_cast<const uint4*>(vbuf + (long long)cell * D + c8 * 8);
    const float ws = expf(sc_s[cell] - ms) * is, wo = expf(sc_o[cell] - mo) * io;
    const uint32_t* u = &pk.x;
    uint4 os, oo;
    uint32_t* ps = &os.x;
    uint32_t* po = &oo.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float lo = __uint_as_float(u[j] << 16), hi = __uint_as_float(u[j] & 0xffff0000u);
      ps[j] = pack_bf16x2(lo * ws, hi * ws);  // model_builder_rel.py:1611
      po[j] = pack_bf16x2(lo * wo, hi * wo);
    }
    const long long o = obase + (long long)cell * D + c8 * 8;
    const long long oso = ((long long)p * cells + cell) * so_stride + c8 * 8;
    *reinterpret_cast<uint4*>(s_map + oso) = os;
    *reinterpret_cast<uint4*>(o_map + oso) = oo;
    *reinterpret_cast<uint4*>(v_map + o) = pk;
  }
}

// ---------------------------------------------------------------------------
// Pair attention + spatio_inferA/B in one pass (model_builder_rel.py:1524-1612, :1039-1041).
// ROIAlign and the 1x1 convs are both linear over the feature map, so they commute:
//   A(s_map) + B(o_map) = a_s (W_A Vp) + a_o (W_B Vp) + b_A + b_B,   W_A Vp = ROIAlign(W_A V-map)
//   q_s . Kp[cell]      = ROIAlign of the scalar map  q_s . K-map
// The caller supplies vab = [V | W_A V | W_B V] (B,H,W,3D) bf16 (one GEMM with host-folded weights) and
// qk (2*n_obj, B*H*W) f32 = [q_s; q_o] K-map^T (row 2i = subject query of object i, 2i+1 = object query).
// Per pair: phase 1 pools the two scalar maps on the union box -> softmax over the cells (a_s, a_o);
// phase 2 pools the 3D channels (a lane owns 8 channels of V, W_A V and W_B V) and writes
//   sp = LeakyReLU(a_s VAp + a_o VBp + bias) + Vp            (P, cells, D) bf16,
// the operand of spatio_infer_fc.  Neither the pooled K-map nor s_map / o_map / v_map reach HBM, and the
// (P*cells) x 2D x D GEMM of spatio_inferA|B disappears.
// ---------------------------------------------------------------------------
struct PSample {   // one bilinear sample of a cell: 4 corner offsets (elements of one map row) + weights
  int pos[4];
  float w[4];
};

__device__ __forceinline__ bool pa_sample(const PGeom& g, int PS, int H, int W, int ph, int pw, int iy, int ix, PSample& sm) {
  float y = g.start_h + (float)ph * g.bin_h + ((float)iy + .5f) * g.bin_h / (float)g.grid_h;
  float x = g.start_w + (float)pw * g.bin_w + ((float)ix + .5f) * g.bin_w / (float)g.grid_w;
  if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return false;
  if (y <= 0.f) y = 0.f;
  if (x <= 0.f) x = 0.f;
  int y_low = (int)y, x_low = (int)x, y_high, x_high;
  float yy = y, xx = x;
  if (y_low >= H - 1) { y_high = y_low = H - 1; yy = (float)y_low; } else { y_high = y_low + 1; }
  if (x_low >= W - 1) { x_high = x_low = W - 1; xx = (float)x_low; } else { x_high = x_low + 1; }
  const float ly = yy - (float)y_low, lx = xx - (float)x_low, hy = 1.f - ly, hx = 1.f - lx;
  sm.w[0] = hy * hx; sm.w[1] = hy * lx; sm.w[2] = ly * hx; sm.w[3] = ly * lx;
  sm.pos[0] = y_low * W + x_low; sm.pos[1] = y_low * W + x_high;
  sm.pos[2] = y_high * W + x_low; sm.pos[3] = y_high * W + x_high;
  return true;
}

__global__ __launch_bounds__(256) void pair_infer_kernel(
    const uint16_t* __restrict__ vab, int B, int H, int W, int D, const float* __restrict__ qk, int ld_qk,
    const float* __restrict__ rel_rois, const int32_t* __restrict__ sbj, const int32_t* __restrict__ obj,
    const int32_t* __restrict__ partner, const float* __restrict__ bias, float scale, int PS, uint16_t* __restrict__ sp) {
  // partner[p] = the pair with subject and object swapped (-1: none).  Both have the same union box, hence the same
  // pooled [V | W_A V | W_B V] cells - 95 % of this kernel's work: the workgroup of the lower index pools once and
  // writes both pairs (each with its own softmax weights), the other one exits.
  __shared__ float sc_s[2][PA_MAX_CELLS], sc_o[2][PA_MAX_CELLS];
  __shared__ float sm_stat[2][4];
  const int p = blockIdx.x;
  const int cells = PS * PS;
  const bool share = partner != nullptr && cells <= 64;
  const int q = share ? partner[p] : -1;
  if (q >= 0 && q < p) return;
  const int npair = q > p ? 2 : 1;
  const int lpb = D >> 3;
  const int bpr = 256 / lpb;
  const int cg = threadIdx.x % lpb;
  const int slot = threadIdx.x / lpb;

  const float* r = rel_rois + (long long)p * 5;
  PGeom g;
  g.batch = min(max((int)r[0], 0), B - 1);
  g.start_w = r[1] * scale;
  g.start_h = r[2] * scale;
  {
    float end_w = r[3] * scale, end_h = r[4] * scale;
    float roi_w = fmaxf(end_w - g.start_w, 1.0f), roi_h = fmaxf(end_h - g.start_h, 1.0f);
    g.bin_h = roi_h / (float)PS;
    g.bin_w = roi_w / (float)PS;
    g.grid_h = (int)ceilf(roi_h / (float)PS);
    g.grid_w = (int)ceilf(roi_w / (float)PS);
  }
  const float inv_count = 1.0f / (float)(g.grid_h * g.grid_w);

  // phase 1: pooled scores of the two scalar maps, one thread per cell (fixed summation order); wave 1 takes the
  // swapped pair
  {
    const int which = threadIdx.x >> 6, cell = threadIdx.x & 63;
    if (which < npair && cell < cells) {
      const int pp = which ? q : p;
      const float* qs_row = qk + (long long)(2 * sbj[pp]) * ld_qk + (long long)g.batch * H * W;
      const float* qo_row = qk + (long long)(2 * obj[pp] + 1) * ld_qk + (long long)g.batch * H * W;
      const int ph = cell / PS, pw = cell - ph * PS;
      float as = 0.f, ao = 0.f;
      for (int iy = 0; iy < g.grid_h; ++iy)
        for (int ix = 0; ix < g.grid_w; ++ix) {
          PSample sm;
          if (!pa_sample(g, PS, H, W, ph, pw, iy, ix, sm)) continue;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            as += sm.w[c] * qs_row[sm.pos[c]];
            ao += sm.w[c] * qo_row[sm.pos[c]];
          }
        }
      sc_s[which][cell] = as * inv_count;
      sc_o[which][cell] = ao * inv_count;
    }
  }
  __syncthreads();
  // softmax statistics over the cells: ONE lane per (pair, map) walks the cells in order and leaves max and 1 / sum in
  // LDS.  (Every lane used to run all four walks for itself: 2 x 2 x cells expf per lane was a third of the kernel's
  // vector work.)  Same order of operations, so the results are unchanged bit for bit.
  if (threadIdx.x < 4) {
    const int k = threadIdx.x >> 1, so = threadIdx.x & 1;
    float mx = -__builtin_inff(), inv = 0.f;
    if (k < npair) {
      const float* sc = so ? sc_o[k] : sc_s[k];
      for (int c = 0; c < cells; ++c) mx = fmaxf(mx, sc[c]);
      float ss = 0.f;
      for (int c = 0; c < cells; ++c) ss += expf(sc[c] - mx);
      inv = 1.0f / ss;
    }
    sm_stat[k][so] = mx;
    sm_stat[k][2 + so] = inv;
  }
  __syncthreads();
  float ms[2], mo[2], is[2], io[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    ms[k] = sm_stat[k][0];
    mo[k] = sm_stat[k][1];
    is[k] = sm_stat[k][2];
    io[k] = sm_stat[k][3];
  }

  // phase 2: pool [V | W_A V | W_B V] and combine
  const uint16_t* img = vab + (long long)g.batch * H * W * (3 * D) + cg * 8;
  float bv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bv[j] = bias[cg * 8 + j];
  for (int cell0 = 0; cell0 < cells; cell0 += bpr) {
    const int cell = cell0 + slot;
    if (cell >= cells) break;
    const int ph = cell / PS, pw = cell - ph * PS;
    float va[8], aa[8], ba[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) va[j] = aa[j] = ba[j] = 0.f;
    for (int iy = 0; iy < g.grid_h; ++iy)
      for (int ix = 0; ix < g.grid_w; ++ix) {
        PSample sm;
        if (!pa_sample(g, PS, H, W, ph, pw, iy, ix, sm)) continue;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const uint16_t* px = img + (long long)sm.pos[c] * (3 * D);
          const uint4 v0 = *reinterpret_cast<const uint4*>(px);
          const uint4 v1 = *reinterpret_cast<const uint4*>(px + D);
          const uint4 v2 = *reinterpret_cast<const uint4*>(px + 2 * D);
          const uint32_t* u0 = &v0.x;
          const uint32_t* u1 = &v1.x;
          const uint32_t* u2 = &v2.x;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            va[2 * j] += sm.w[c] * __uint_as_float(u0[j] << 16);
            va[2 * j + 1] += sm.w[c] * __uint_as_float(u0[j] & 0xffff0000u);
            aa[2 * j] += sm.w[c] * __uint_as_float(u1[j] << 16);
            aa[2 * j + 1] += sm.w[c] * __uint_as_float(u1[j] & 0xffff0000u);
            ba[2 * j] += sm.w[c] * __uint_as_float(u2[j] << 16);
            ba[2 * j + 1] += sm.w[c] * __uint_as_float(u2[j] & 0xffff0000u);
          }
        }
      }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (k >= npair) break;
      const float ws = expf(sc_s[k][cell] - ms[k]) * is[k] * inv_count, wo = expf(sc_o[k][cell] - mo[k]) * io[k] * inv_count;
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float t = ws * aa[j] + wo * ba[j] + bv[j];       // A(s_map) + B(o_map)  (:1039)
        t = t > 0.f ? t : 0.1f * t;                      // LeakyReLU(0.1)
        o[j] = t + va[j] * inv_count;                    // + V  (:1040)
      }
      uint4 pk;
      pk.x = pack_bf16x2(o[0], o[1]);
      pk.y = pack_bf16x2(o[2], o[3]);
      pk.z = pack_bf16x2(o[4], o[5]);
      pk.w = pack_bf16x2(o[6], o[7]);
      *reinterpret_cast<uint4*>(sp + ((long long)(k ? q : p) * cells + cell) * D + cg * 8) = pk;
    }
  }
}

// ---------------------------------------------------------------------------
// The same computation as pair_infer_kernel in two launches, so that the heavy half leaves the critical
// path: pair_pool_kernel (union-box pooling of [V | W_A V | W_B V]) needs only the 2-D feature map and the
// pair boxes and runs beside the I3D backbone on the second stream; pair_combine_kernel (scores of the
// scalar maps q.K, softmax over the cells, LeakyReLU(a_s VAp + a_o VBp + b) + Vp) is what remains after
// the tree-GRU has produced the queries: a streaming pass over the pooled maps.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void pa_geometry(const float* r, int B, float scale, int PS, PGeom& g) {
  g.batch = min(max((int)r[0], 0), B - 1);
  g.start_w = r[1] * scale;
  g.start_h = r[2] * scale;
  const float end_w = r[3] * scale, end_h = r[4] * scale;
  const float roi_w = fmaxf(end_w - g.start_w, 1.0f), roi_h = fmaxf(end_h - g.start_h, 1.0f);
  g.bin_h = roi_h / (float)PS;
  g.bin_w = roi_w / (float)PS;
  g.grid_h = (int)ceilf(roi_h / (float)PS);
  g.grid_w = (int)ceilf(roi_w / (float)PS);
}

__global__ __launch_bounds__(256) void pair_pool_kernel(const uint16_t* __restrict__ vab, int B, int H, int W, int D,
                                                        const float* __restrict__ rel_rois,
                                                        const int32_t* __restrict__ partner, float scale, int PS,
                                                        uint16_t* __restrict__ pooled) {
  const int p = blockIdx.x;
  if (partner) {   // swapped pairs share a union box: only the lower index pools (row p of `pooled`)
    const int q = partner[p];
    if (q >= 0 && q < p) return;
  }
  const int cells = PS * PS;
  const int lpb = D >> 3, bpr = 256 / lpb;
  const int cg = threadIdx.x % lpb, slot = threadIdx.x / lpb;
  PGeom g;
  pa_geometry(rel_rois + (long long)p * 5, B, scale, PS, g);
  const float inv_count = 1.0f / (float)(g.grid_h * g.grid_w);
  const uint16_t* img = vab + (long long)g.batch * H * W * (3 * D) + cg * 8;
  for (int cell0 = 0; cell0 < cells; cell0 += bpr) {
    const int cell = cell0 + slot;
    if (cell >= cells) break;
    const int ph = cell / PS, pw = cell - ph * PS;
    float acc[3][8];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[k][j] = 0.f;
    for (int iy = 0; iy < g.grid_h; ++iy)
      for (int ix = 0; ix < g.grid_w; ++ix) {
        PSample sm;
        if (!pa_sample(g, PS, H, W, ph, pw, iy, ix, sm)) continue;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const uint16_t* px = img + (long long)sm.pos[c] * (3 * D);
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const uint4 v = *reinterpret_cast<const uint4*>(px + k * D);
            const uint32_t* u = &v.x;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              acc[k][2 * j] += sm.w[c] * __uint_as_float(u[j] << 16);
              acc[k][2 * j + 1] += sm.w[c] * __uint_as_float(u[j] & 0xffff0000u);
            }
          }
        }
      }
    uint16_t* o = pooled + ((long long)p * cells + cell) * (3 * D) + cg * 8;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      uint4 pk;
      pk.x = pack_bf16x2(acc[k][0] * inv_count, acc[k][1] * inv_count);
      pk.y = pack_bf16x2(acc[k][2] * inv_count, acc[k][3] * inv_count);
      pk.z = pack_bf16x2(acc[k][4] * inv_count, acc[k][5] * inv_count);
      pk.w = pack_bf16x2(acc[k][6] * inv_count, acc[k][7] * inv_count);
      *reinterpret_cast<uint4*>(o + k * D) = pk;
    }
  }
}

__global__ __launch_bounds__(256) void pair_combine_kernel(
    const uint16_t* __restrict__ pooled, int B, int H, int W, int D, const float* __restrict__ qk, int ld_qk,
    const float* __restrict__ rel_rois, const int32_t* __restrict__ sbj, const int32_t* __restrict__ obj,
    const int32_t* __restrict__ partner, const float* __restrict__ bias, float scale, int PS, uint16_t* __restrict__ sp) {
  // with `partner`: the lower index of a swapped pair reads the shared pooled row ONCE and writes both rows of sp
  __shared__ float sc_s[2][PA_MAX_CELLS], sc_o[2][PA_MAX_CELLS];
  __shared__ float sm_stat[2][4];
  const int p = blockIdx.x;
  const int cells = PS * PS;
  const bool share = partner != nullptr && cells <= 64;
  int q = partner ? partner[p] : -1;
  int prow = p;            // row of `pooled` that holds this pair's union-box pooling
  if (q >= 0 && q < p) {
    if (share) return;
    prow = q;
  }
  const int npair = (share && q > p) ? 2 : 1;
  const int lpb = D >> 3, bpr = 256 / lpb;
  const int cg = threadIdx.x % lpb, slot = threadIdx.x / lpb;
  PGeom g;
  pa_geometry(rel_rois + (long long)p * 5, B, scale, PS, g);
  const float inv_count = 1.0f / (float)(g.grid_h * g.grid_w);
  {   // same summation order as pair_infer_kernel; wave 1 takes the swapped pair
    const int which = threadIdx.x >> 6, cell = threadIdx.x & 63;
    if (which < npair && cell < cells) {
      const int pp = which ? q : p;
      const float* qs_row = qk + (long long)(2 * sbj[pp]) * ld_qk + (long long)g.batch * H * W;
      const float* qo_row = qk + (long long)(2 * obj[pp] + 1) * ld_qk + (long long)g.batch * H * W;
      const int ph = cell / PS, pw = cell - ph * PS;
      float as = 0.f, ao = 0.f;
      for (int iy = 0; iy < g.grid_h; ++iy)
        for (int ix = 0; ix < g.grid_w; ++ix) {
          PSample sm;
          if (!pa_sample(g, PS, H, W, ph, pw, iy, ix, sm)) continue;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            as += sm.w[c] * qs_row[sm.pos[c]];
            ao += sm.w[c] * qo_row[sm.pos[c]];
          }
        }
      sc_s[which][cell] = as * inv_count;
      sc_o[which][cell] = ao * inv_count;
    }
  }
  __syncthreads();
  // softmax statistics over the cells: ONE lane per (pair, map) walks the cells in order and leaves max and 1 / sum in
  // LDS.  (Every lane used to run all four walks for itself: 2 x 2 x cells expf per lane was a third of the kernel's
  // vector work.)  Same order of operations, so the results are unchanged bit for bit.
  if (threadIdx.x < 4) {
    const int k = threadIdx.x >> 1, so = threadIdx.x & 1;
    float mx = -__builtin_inff(), inv = 0.f;
    if (k < npair) {
      const float* sc = so ? sc_o[k] : sc_s[k];
      for (int c = 0; c < cells; ++c) mx = fmaxf(mx, sc[c]);
      float ss = 0.f;
      for (int c = 0; c < cells; ++c) ss += expf(sc[c] - mx);
      inv = 1.0f / ss;
    }
    sm_stat[k][so] = mx;
    sm_stat[k][2 + so] = inv;
  }
  __syncthreads();
  float ms[2], mo[2], is[2], io[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    ms[k] = sm_stat[k][0];
    mo[k] = sm_stat[k][1];
    is[k] = sm_stat[k][2];
    io[k] = sm_stat[k][3];
  }
  float bv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bv[j] = bias[cg * 8 + j];
  for (int cell0 = 0; cell0 < cells; cell0 += bpr) {
    const int cell = cell0 + slot;
    if (cell >= cells) break;
    const uint16_t* src = pooled + ((long long)prow * cells + cell) * (3 * D) + cg * 8;
    const uint4 v0 = *reinterpret_cast<const uint4*>(src);
    const uint4 v1 = *reinterpret_cast<const uint4*>(src + D);
    const uint4 v2 = *reinterpret_cast<const uint4*>(src + 2 * D);
    const uint32_t* u0 = &v0.x;
    const uint32_t* u1 = &v1.x;
    const uint32_t* u2 = &v2.x;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (k >= npair) break;
      const float ws = expf(sc_s[k][cell] - ms[k]) * is[k], wo = expf(sc_o[k][cell] - mo[k]) * io[k];
      float o[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float vl = __uint_as_float(u0[j] << 16), vh = __uint_as_float(u0[j] & 0xffff0000u);
        const float al = __uint_as_float(u1[j] << 16), ah = __uint_as_float(u1[j] & 0xffff0000u);
        const float bl = __uint_as_float(u2[j] << 16), bh = __uint_as_float(u2[j] & 0xffff0000u);
        float t0 = ws * al + wo * bl + bv[2 * j], t1 = ws * ah + wo * bh + bv[2 * j + 1];
        t0 = t0 > 0.f ? t0 : 0.1f * t0;
        t1 = t1 > 0.f ? t1 : 0.1f * t1;
        o[2 * j] = t0 + vl;
        o[2 * j + 1] = t1 + vh;
      }
      uint4 pk;
      pk.x = pack_bf16x2(o[0], o[1]);
      pk.y = pack_bf16x2(o[2], o[3]);
      pk.z = pack_bf16x2(o[4], o[5]);
      pk.w = pack_bf16x2(o[6], o[7]);
      *reinterpret_cast<uint4*>(sp + ((long long)(k ? q : p) * cells + cell) * D + cg * 8) = pk;
    }
  }
}

// ---------------------------------------------------------------------------
// Pair attention as MFMA tiles (model_builder_rel.py:1524-1612, :1039-1041; the arithmetic of pair_infer_kernel).
// The union-box pooling of a map is a small matrix product over the map's pixels,
//     pooled[ch][cell] = sum_pix mapT[ch][pix] * Wt[pix][cell],   Wt[(y,x)][(cy,cx)] = wy[cy][y] * wx[cx][x]
// (wy / wx: the bilinear weights of the cell's samples along one axis, summed and divided by the sample count; a
// sample is dropped iff its y OR its x is out of range, so the product form is exact), and one workgroup computes it for
// all 3 D channels of [V | W_A V | W_B V] of one union box on v_mfma_f32_16x16x32_bf16:
//   * the maps arrive CHANNEL-major, vabT (B, 3D, ldp) bf16 with the pixel axis contiguous (the producing GEMM simply
//     runs transposed), so the A operand (rows = channels, K = pixels) is a 16-byte global load per lane - no LDS
//     staging, no transpose.  K runs over the box's pixel ROWS (y0 W .. (y1 + 1) W, from an 8-aligned start);
//   * the B operand is the weight image Wt (bf16, [64 cells][K]) built in LDS from the two 7 x 14 axis tables;
//   * scores of the scalar maps q.K (f32, from the same axis tables), softmax over the cells, and the combine
//     sp = LeakyReLU(a_s VAp + a_o VBp + b) + Vp run on the accumulators; the map GEMM's bias enters as
//     bias[ch] * (sum of the cell's weights) - the pooled value of a constant map;
//   * the swapped pair (same union box) is written from the same accumulators.
// Eight waves: wave w owns channels 32 w .. 32 w + 31 of each of the three maps (rows permuted so that a lane ends up
// with 8 consecutive channels of a cell -> 16-byte stores) and all 64 cell columns: 24 accumulator tiles.
// ---------------------------------------------------------------------------
constexpr int PM_LDP = 224;        // pixel pitch of vabT (>= H * W rounded up to 8, + 24)
constexpr int PM_WROW = PM_LDP + 48; // weight image row (elements): 544 B = 32 B mod the 256-B bank row, so the 16 lanes of a ds_read_b128 group (8 cells of one k quad + 8 of the next) fall on 16 different slots
constexpr int PM_THREADS = 256;
typedef __attribute__((ext_vector_type(8))) short pm_s8;
typedef __attribute__((ext_vector_type(4))) float pm_f4;

__global__ __launch_bounds__(PM_THREADS, 2) void pair_mfma_kernel(
    const uint16_t* __restrict__ vabT, int B, int H, int W, int D, const float* __restrict__ vbias,
    const float* __restrict__ qk, int ld_qk, const float* __restrict__ rel_rois, const int32_t* __restrict__ sbj,
    const int32_t* __restrict__ obj, const int32_t* __restrict__ partner, const int32_t* __restrict__ work,
    const float* __restrict__ bias_ab, float scale, int PS, uint16_t* __restrict__ sp) {
  __shared__ __attribute__((aligned(16))) uint16_t wimg[64 * PM_WROW];
  __shared__ float wy[8][16], wx[8][16], sy[8], sx[8];
  __shared__ int ylo[8], yhi[8], xlo[8], xhi[8];   // pixels with a non-zero weight, per bin and axis (lo > hi: none)
  __shared__ float qrow[4][PM_LDP];
  __shared__ float prob[4][64];        // pooled scores, then softmax weights: [2 pair + (s | o)][cell]
  __shared__ __attribute__((aligned(16))) float bias_l[4][512];   // map bias of V | W_A V | W_B V, then b_A + b_B  (D <= 512)
  const int tid = threadIdx.x;
  const int cells = PS * PS;
  const int HW = H * W;
  // One workgroup per union box.  With `work` (12 words per box: pair p, swapped pair q or -1, subject and object of p,
  // the box, padding) everything the box needs arrives in one 48-byte read and no workgroup is launched for the swapped
  // pairs; without it a workgroup per pair looks its inputs up and the higher index of a swapped pair exits.
  int p, q, s_i, o_i;
  float roi[5];
  if (work) {
    const int4 w0 = *reinterpret_cast<const int4*>(work + 12 * (long long)blockIdx.x);
    const float4 w1 = *reinterpret_cast<const float4*>(work + 12 * (long long)blockIdx.x + 4);
    p = w0.x; q = w0.y; s_i = w0.z; o_i = w0.w;
    roi[0] = w1.x; roi[1] = w1.y; roi[2] = w1.z; roi[3] = w1.w;
    roi[4] = __int_as_float(work[12 * (long long)blockIdx.x + 8]);
  } else {
    p = blockIdx.x;
    q = partner ? partner[p] : -1;
    if (q >= 0 && q < p) return;
    s_i = sbj[p];
    o_i = obj[p];
#pragma unroll
    for (int i = 0; i < 5; ++i) roi[i] = rel_rois[(long long)p * 5 + i];
  }
  const int npair = q > p ? 2 : 1;

  PGeom g;
  pa_geometry(roi, B, scale, PS, g);
  // ---- axis tables: thread (axis, bin) walks its samples (model_builder_rel.py:1566-1569 -> ROIAlign_cuda.cu:64-122)
  if (tid < 2 * PS) {
    const int axis = tid >= PS, b = axis ? tid - PS : tid;
    const float start = axis ? g.start_w : g.start_h, bin = axis ? g.bin_w : g.bin_h;
    const int grid = axis ? g.grid_w : g.grid_h, dim = axis ? W : H;
    float* row = axis ? wx[b] : wy[b];
#pragma unroll
    for (int i = 0; i < 16; ++i) row[i] = 0.f;
    const float inv = 1.0f / (float)grid;
    float tot = 0.f;
    int first = 16, last = -1;
    for (int i = 0; i < grid; ++i) {
      float c = start + (float)b * bin + ((float)i + .5f) * bin / (float)grid;
      if (c < -1.0f || c > (float)dim) continue;
      if (c <= 0.f) c = 0.f;
      int lo = (int)c, hi;
      if (lo >= dim - 1) { hi = lo = dim - 1; c = (float)lo; } else { hi = lo + 1; }
      const float l = c - (float)lo, h = 1.f - l;
      row[lo] += h * inv;
      row[hi] += l * inv;
      tot += inv;
      first = min(first, lo);
      last = max(last, hi);
    }
    (axis ? sx : sy)[b] = tot;
    (axis ? xlo : ylo)[b] = first;
    (axis ? xhi : yhi)[b] = last;
  }
  // the weight image starts as zeros: a cell touches at most (grid + 1)^2 of its <= 224 pixels
  {
    const int cell = tid & 63;
    for (int c8 = tid >> 6; c8 < PM_LDP / 8; c8 += PM_THREADS / 64)
      *reinterpret_cast<uint4*>(&wimg[cell * PM_WROW + 8 * c8]) = make_uint4(0, 0, 0, 0);
  }
  for (int i = tid; i < 4 * D; i += PM_THREADS) {
    const int mt = i / D, c = i - mt * D;
    bias_l[mt][c] = mt < 3 ? (vbias ? vbias[i] : 0.f) : bias_ab[c];
  }
  // ---- the queries' scalar maps of this clip (rows 2 i = q_s of object i, 2 i + 1 = q_o), zero behind the map
  {
    const int which = tid >> 6, l = tid & 63;
    if (which < 2 * npair) {
      // the swapped pair's subject is this pair's object and vice versa
      const int rowi = (which & 1) ? 2 * ((which >> 1) ? s_i : o_i) + 1 : 2 * ((which >> 1) ? o_i : s_i);
      const float* src = qk + (long long)rowi * ld_qk + (long long)g.batch * HW;
#pragma unroll
      for (int i = 0; i < (PM_LDP + 63) / 64; ++i) {
        const int px = l + 64 * i;
        if (px < PM_LDP) qrow[which][px] = px < HW ? src[px] : 0.f;
      }
    }
  }
  __syncthreads();
  // ---- pixel rows the cells touch (uniform)
  int ymin = 16, ymax = -1;
#pragma unroll
  for (int b = 0; b < 8; ++b)
    if (b < PS) {
      ymin = min(ymin, ylo[b]);
      ymax = max(ymax, yhi[b]);
    }
  const int k0 = ymax < 0 ? 0 : ((ymin * W) & ~7);
  const int ksteps = ymax < 0 ? 0 : ((ymax + 1) * W - k0 + 31) >> 5;
  // ---- the first K-step's A fragments of this wave's first channel group: in flight while the image is built
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kg = lane >> 4;
  const uint16_t* arow[3][2];
  pm_s8 af[2][3][2];
  auto set_rows = [&](int cb) {
    // A rows: tile t of the wave's pair of 16-row tiles holds channels cb + 8 (m >> 2) + 4 t + (m & 3), m = li
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int ch = cb + 8 * (li >> 2) + 4 * t + (li & 3);
        arow[mt][t] = vabT + ((long long)(g.batch * 3 + mt) * D + ch) * PM_LDP + k0 + 8 * kg;
      }
  };
  auto load_step = [&](int set, int ks) {
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
      for (int t = 0; t < 2; ++t) af[set][mt][t] = *reinterpret_cast<const pm_s8*>(arow[mt][t] + 32 * ks);
  };
  if (wave * 32 < D) {
    set_rows(wave * 32);
    if (ksteps > 0) load_step(0, 0);
  }
  // ---- weight image [cell][pixel - k0] (bf16) and pooled scores, both over the cell's non-zero pixels only:
  // thread (part = tid >> 6, cell = tid & 63).  Image: part takes every fourth of the cell's pixel rows.  Scores: part =
  // which (2 pair + (s | o)), then the softmax over the cells with lane = cell.
  {
    const int part = tid >> 6, cell = tid & 63;
    float a = -__builtin_inff();
    if (cell < cells) {
      const int cy = cell / PS, cx = cell - cy * PS;
      const int y0 = ylo[cy], y1 = yhi[cy], x0 = xlo[cx], x1 = xhi[cx];
      for (int y = y0 + part; y <= y1; y += PM_THREADS / 64) {
        const float wyv = wy[cy][y];
        uint16_t* dst = &wimg[cell * PM_WROW + y * W - k0];
        for (int x = x0; x <= x1; ++x) dst[x] = f32_to_bf16(wyv * wx[cx][x]);
      }
      if (part < 2 * npair) {
        a = 0.f;
        for (int y = y0; y <= y1; ++y) {
          const float* qr = &qrow[part][y * W];
          float r = 0.f;
          for (int x = x0; x <= x1; ++x) r += wx[cx][x] * qr[x];
          a += wy[cy][y] * r;
        }
      }
    }
    if (part < 2 * npair) {
      float mx = a;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
      const float e = cell < cells ? expf(a - mx) : 0.f;
      float ss = e;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
      prob[part][cell] = e / ss;
    }
  }
  __syncthreads();

  for (int cb = wave * 32; cb < D; cb += (PM_THREADS / 64) * 32) {
    pm_f4 acc[3][2][4];
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) acc[mt][t][nf] = pm_f4{0.f, 0.f, 0.f, 0.f};
    // the NEXT K-step's fragments are loaded before this step's 24 MFMAs (two register sets): an L2 round trip hides
    // behind the matrix work of this wave and of the other workgroup on the CU; step 0 of a channel group is loaded
    // before the previous group's epilogue (before the image build for the first group)
    auto mma_step = [&](int set, int ks) {
      pm_s8 bf[4];
#pragma unroll
      for (int nf = 0; nf < 4; ++nf)
        bf[nf] = *reinterpret_cast<const pm_s8*>(&wimg[(16 * nf + li) * PM_WROW + 32 * ks + 8 * kg]);
#pragma unroll
      for (int nf = 0; nf < 4; ++nf)
#pragma unroll
        for (int mt = 0; mt < 3; ++mt)
#pragma unroll
          for (int t = 0; t < 2; ++t)
            acc[mt][t][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[set][mt][t], bf[nf], acc[mt][t][nf], 0, 0, 0);
    };
    for (int ks = 0; ks < ksteps; ks += 2) {
      if (ks + 1 < ksteps) load_step(1, ks + 1);
      mma_step(0, ks);
      if (ks + 1 < ksteps) {
        if (ks + 2 < ksteps) load_step(0, ks + 2);
        mma_step(1, ks + 1);
      }
    }
    if (cb + (PM_THREADS / 64) * 32 < D) {
      set_rows(cb + (PM_THREADS / 64) * 32);
      if (ksteps > 0) load_step(0, 0);
    }
    // ---- combine: the lane holds channels c .. c + 7 (rows 4 kg + r of tiles t = 0, 1) of cell 16 nf + li
    const int c = cb + 8 * kg;
    float vb[3][8], bab[8];
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
      for (int j = 0; j < 8; ++j) vb[mt][j] = bias_l[mt][c + j];
#pragma unroll
    for (int j = 0; j < 8; ++j) bab[j] = bias_l[3][c + j];
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) {
      const int cell = 16 * nf + li;
      if (cell >= cells) continue;
      const int cy = cell / PS, cx = cell - cy * PS;
      const float wsum = sy[cy] * sx[cx];
      float v[8], a[8], b[8];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[4 * t + r] = acc[0][t][nf][r] + vb[0][4 * t + r] * wsum;
          a[4 * t + r] = acc[1][t][nf][r] + vb[1][4 * t + r] * wsum;
          b[4 * t + r] = acc[2][t][nf][r] + vb[2][4 * t + r] * wsum;
        }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        if (k >= npair) break;
        const float ws = prob[2 * k][cell], wo = prob[2 * k + 1][cell];
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float t = ws * a[j] + wo * b[j] + bab[j];       // A(s_map) + B(o_map)  (:1039)
          t = t > 0.f ? t : 0.1f * t;                     // LeakyReLU(0.1)
          o[j] = t + v[j];                                // + V  (:1040)
        }
        uint4 pk;
        pk.x = pack_bf16x2(o[0], o[1]);
        pk.y = pack_bf16x2(o[2], o[3]);
        pk.z = pack_bf16x2(o[4], o[5]);
        pk.w = pack_bf16x2(o[6], o[7]);
        *reinterpret_cast<uint4*>(sp + ((long long)(k ? q : p) * cells + cell) * D + c) = pk;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Tree-GRU: one workgroup per (tree, feature group); levels are walked inside
// the kernel (a tree is independent of every other tree), so the whole
// bottom-up + top-down pass of all clips and groups is ONE launch.
// Weight matrices are passed TRANSPOSED ((in, out) row-major) so that lane j
// reads consecutive addresses.
// ---------------------------------------------------------------------------
struct TreeW {
  const float *Wz, *Wr, *Wh, *Uz, *Ur, *Uh, *bz, *br;
  const float *Wzd, *Wrd, *Whd, *Uzd, *Urd, *Uhd, *bzd, *brd;
};
struct TreeParams {
  const float* x;
  int F, Hd, G;
  const int32_t *tree_off, *level_off, *n_levels, *parent, *child_off, *child_idx;
  float *h, *hd, *ws;
  TreeW w[TRACE_TREE_MAX_GROUPS];
};

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

// dot of column j of a transposed (n_in, Hd) matrix (LDS or global) with a contiguous vector;
// n_in % 4 == 0 (checked by the host wrapper): 16-byte broadcast loads of v, 4 partial sums.
__device__ __forceinline__ float dot_col(const float* __restrict__ Wt, int Hd, int j,
                                         const float* __restrict__ v, int n_in) {
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  for (int i = 0; i < n_in; i += 4) {
    const float4 x = *reinterpret_cast<const float4*>(v + i);
    s0 += Wt[(i + 0) * Hd + j] * x.x;
    s1 += Wt[(i + 1) * Hd + j] * x.y;
    s2 += Wt[(i + 2) * Hd + j] * x.z;
    s3 += Wt[(i + 3) * Hd + j] * x.w;
  }
  return (s0 + s1) + (s2 + s3);
}

// LDS = true: the six weight matrices of the current direction are staged in LDS (lane j reads
// consecutive banks, the vector element is a broadcast); false: read through L1/L2 (large Hd).
template <bool LDS>
__global__ __launch_bounds__(256) void tree_gru_kernel(const TreeParams p) {
  extern __shared__ __attribute__((aligned(16))) float wlds[];
  const int t = blockIdx.x, g = blockIdx.y;
  const TreeW& w = p.w[g];
  const int F = p.F, Hd = p.Hd;
  const long long ldx = (long long)p.G * F, ldh = (long long)p.G * Hd;
  const float* x = p.x + (long long)g * F;
  float* h = p.h + (long long)g * Hd;
  float* hd = p.hd + (long long)g * Hd;
  float* ws = p.ws + (long long)g * Hd;
  const int nl = p.n_levels[t];
  const int32_t* lo = p.level_off + (long long)t * (TRACE_TREE_MAX_LEVELS + 1);

  const float *Wz = w.Wz, *Wr = w.Wr, *Wh = w.Wh, *Uz = w.Uz, *Ur = w.Ur, *Uh = w.Uh;
  auto stage = [&](const float* a0, const float* a1, const float* a2, const float* b0, const float* b1,
                   const float* b2) {
    if (!LDS) {
      Wz = a0; Wr = a1; Wh = a2; Uz = b0; Ur = b1; Uh = b2;
      return;
    }
    __syncthreads();
    const int nw = F * Hd, nu = Hd * Hd;
    float* d = wlds;
    const float* srcs[6] = {a0, a1, a2, b0, b1, b2};
    for (int k = 0; k < 6; ++k) {
      const int n = k < 3 ? nw : nu;
      for (int i = threadIdx.x * 4; i < n; i += blockDim.x * 4)
        *reinterpret_cast<float4*>(d + i) = *reinterpret_cast<const float4*>(srcs[k] + i);
      d += n;
    }
    Wz = wlds; Wr = Wz + nw; Wh = Wr + nw; Uz = Wh + nw; Ur = Uz + nu; Uh = Ur + nu;
    __syncthreads();
  };
  stage(w.Wz, w.Wr, w.Wh, w.Uz, w.Ur, w.Uh);

  // ---- bottom-up, leaves (cluster_tree.py:289-292)
  {
    const int v0 = lo[0], cnt = (lo[1] - lo[0]) * Hd;
    for (int idx = threadIdx.x; idx < cnt; idx += blockDim.x) {
      const int v = v0 + idx / Hd, j = idx % Hd;
      const float* xv = x + v * ldx;
      float z = sigmoidf_(dot_col(Wz, Hd, j, xv, F));
      float c = tanhf(dot_col(Wh, Hd, j, xv, F));
      h[v * ldh + j] = (1.0f - z) * c;
    }
  }
  __threadfence_block();
  __syncthreads();
  // ---- bottom-up, inner levels (cluster_tree.py:294-315)
  for (int l = 1; l < nl; ++l) {
    const int v0 = lo[l], cnt = (lo[l + 1] - lo[l]) * Hd;
    // phase A: fa_g[v][j] = sum_c sigmoid(Wr x + Ur h_c + br)[j] * h_c[j]
    for (int idx = threadIdx.x; idx < cnt; idx += blockDim.x) {
      const int v = v0 + idx / Hd, j = idx % Hd;
      const float a = dot_col(Wr, Hd, j, x + v * ldx, F) + w.br[j];
      float fa = 0.f;
      for (int ci = p.child_off[v]; ci < p.child_off[v + 1]; ++ci) {
        const float* hc = h + (long long)p.child_idx[ci] * ldh;
        fa += sigmoidf_(a + dot_col(Ur, Hd, j, hc, Hd)) * hc[j];
      }
      ws[v * ldh + j] = fa;
    }
    __threadfence_block();
    __syncthreads();
    // phase B: z, candidate, h
    for (int idx = threadIdx.x; idx < cnt; idx += blockDim.x) {
      const int v = v0 + idx / Hd, j = idx % Hd;
      const float* xv = x + v * ldx;
      float az = dot_col(Wz, Hd, j, xv, F) + w.bz[j];
      float hs = 0.f;
      for (int ci = p.child_off[v]; ci < p.child_off[v + 1]; ++ci) {
        const float* hc = h + (long long)p.child_idx[ci] * ldh;
        az += dot_col(Uz, Hd, j, hc, Hd);
        hs += hc[j];
      }
      const float z = sigmoidf_(az);
      const float c = tanhf(dot_col(Wh, Hd, j, xv, F) + dot_col(Uh, Hd, j, ws + v * ldh, Hd));
      h[v * ldh + j] = z * hs + (1.0f - z) * c;
    }
    __threadfence_block();
    __syncthreads();
  }
  // ---- top-down (cluster_tree.py:189-200, 326-330): roots copy h
  stage(w.Wzd, w.Wrd, w.Whd, w.Uzd, w.Urd, w.Uhd);
  {
    const int v0 = lo[nl - 1], cnt = (lo[nl] - lo[nl - 1]) * Hd;
    for (int idx = threadIdx.x; idx < cnt; idx += blockDim.x) {
      const int v = v0 + idx / Hd, j = idx % Hd;
      hd[v * ldh + j] = h[v * ldh + j];
    }
  }
  __threadfence_block();
  __syncthreads();
  for (int l = nl - 2; l >= 0; --l) {
    const int v0 = lo[l], cnt = (lo[l + 1] - lo[l]) * Hd;
    // phase A: rh[v][j] = sigmoid(Wrd x + Urd hd_f + brd)[j] * hd_f[j]
    for (int idx = threadIdx.x; idx < cnt; idx += blockDim.x) {
      const int v = v0 + idx / Hd, j = idx % Hd;
      const int f = p.parent[v];
      const float* hf = hd + (long long)f * ldh;
      float r = sigmoidf_(dot_col(Wr, Hd, j, x + v * ldx, F) + dot_col(Ur, Hd, j, hf, Hd) + w.brd[j]);
      ws[v * ldh + j] = r * hf[j];
    }
    __threadfence_block();
    __syncthreads();
    for (int idx = threadIdx.x; idx < cnt; idx += blockDim.x) {
      const int v = v0 + idx / Hd, j = idx % Hd;
      const int f = p.parent[v];
      const float* hf = hd + (long long)f * ldh;
      const float* xv = x + v * ldx;
      float z = sigmoidf_(dot_col(Wz, Hd, j, xv, F) + dot_col(Uz, Hd, j, hf, Hd) + w.bzd[j]);
      float c = tanhf(dot_col(Wh, Hd, j, xv, F) + dot_col(Uh, Hd, j, ws + v * ldh, Hd));
      hd[v * ldh + j] = z * hf[j] + (1.0f - z) * c;
    }
    __threadfence_block();
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// Tree-GRU, LDS-resident variant (used when one tree's state and one direction's matrices fit in
// LDS: the VidVRD head, F = Hd = 64, <= ~70 nodes per tree).  The pass is a chain of ~2*levels
// dependent steps, so every operand of the chain lives in LDS:
//   * per direction, the input terms W{z,r,h} x_v of ALL nodes are computed first (no dependence
//     between nodes), from LDS-staged W and x rows;
//   * then the three U matrices replace W in the same LDS region and the levels are walked with
//     h / hd / the gated sums read and written in LDS only; h and hd go to HBM once at the end;
//   * U_z applied to the sum of the children states equals the sum of U_z h_c (linear), which saves
//     one matrix-vector product per extra child.
// dot products: lane j owns output j; column j of a transposed matrix is one LDS bank, the vector
// element is a broadcast.
// ---------------------------------------------------------------------------
// CN > 0: compile-time length (fully unrolled: all LDS reads of a product are issued back to back
// instead of one exposed LDS latency per 4 MACs)
template <int CN>
__device__ __forceinline__ float dot_lds(const float* __restrict__ Wt, int Hd, int j,
                                         const float* __restrict__ v, int n_in) {
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  const int n = CN > 0 ? CN : n_in;
#pragma unroll
  for (int i = 0; i < n; i += 4) {
    const float4 x = *reinterpret_cast<const float4*>(v + i);
    s0 += Wt[(i + 0) * Hd + j] * x.x;
    s1 += Wt[(i + 1) * Hd + j] * x.y;
    s2 += Wt[(i + 2) * Hd + j] * x.z;
    s3 += Wt[(i + 3) * Hd + j] * x.w;
  }
  return (s0 + s1) + (s2 + s3);
}

template <int CF, int CH>
__global__ __launch_bounds__(512) void tree_gru_lds_kernel(const TreeParams p, int nn_max) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int t = blockIdx.x, g = blockIdx.y;
  const TreeW& w = p.w[g];
  const int F = CF > 0 ? CF : p.F, Hd = CH > 0 ? CH : p.Hd;
  const int FH = F > Hd ? F : Hd;
  float* R = sm;                       // 3 matrices of one kind: [3][FH][Hd]
  float* xt = R + 3 * FH * Hd;         // input terms of the current direction [3][nn_max][Hd]
  float* hl = xt + 3 * nn_max * Hd;    // bottom-up state   [nn_max][Hd]
  float* hdl = hl + nn_max * Hd;       // top-down state    [nn_max][Hd]
  float* wsl = hdl + nn_max * Hd;      // gated sums [nn_max][Hd]; first holds the x rows [nn_max][F]
  float* bias = wsl + nn_max * FH;     // bz, br, bzd, brd [4][Hd]
  int* par_l = (int*)(bias + 4 * Hd);  // parent (local id) [nn_max]
  int* coff_l = par_l + nn_max;        // child list offsets (local) [nn_max + 1]
  int* cidx_l = coff_l + nn_max + 1;   // children (local ids) [nn_max]
  __shared__ int lo_s[TRACE_TREE_MAX_LEVELS + 2];
  const long long ldx = (long long)p.G * F, ldh = (long long)p.G * Hd;
  const int vb = p.tree_off[t];
  const int nn = p.tree_off[t + 1] - vb;
  float* hg = p.h + (long long)g * Hd;
  float* hdg = p.hd + (long long)g * Hd;
  if (nn > nn_max) {  // caller's bound was wrong: poison this tree's outputs instead of overrunning LDS
    for (int idx = threadIdx.x; idx < nn * Hd; idx += blockDim.x) {
      hg[(vb + idx / Hd) * ldh + idx % Hd] = __builtin_nanf("");
      hdg[(vb + idx / Hd) * ldh + idx % Hd] = __builtin_nanf("");
    }
    return;
  }
  const int nl = p.n_levels[t];
  const float* xg = p.x + (long long)g * F;
  // the tree's topology and the gate biases are read on every step of the dependent chain: keep them
  // in LDS (a global load per level would put an L2 round trip on the critical path each time)
  {
    const int32_t* lo_g = p.level_off + (long long)t * (TRACE_TREE_MAX_LEVELS + 1);
    const int cbase = p.child_off[vb];
    for (int i = threadIdx.x; i <= TRACE_TREE_MAX_LEVELS; i += blockDim.x) lo_s[i] = lo_g[i] - vb;
    for (int i = threadIdx.x; i < nn; i += blockDim.x) {
      const int pg = p.parent[vb + i];
      par_l[i] = pg < 0 ? -1 : pg - vb;
    }
    for (int i = threadIdx.x; i <= nn; i += blockDim.x) coff_l[i] = p.child_off[vb + i] - cbase;
    const int nch = p.child_off[vb + nn] - cbase;   // <= nn - 1 edges
    for (int i = threadIdx.x; i < nch && i < nn_max; i += blockDim.x) cidx_l[i] = p.child_idx[cbase + i] - vb;
    for (int i = threadIdx.x; i < Hd; i += blockDim.x) {
      bias[i] = w.bz[i];
      bias[Hd + i] = w.br[i];
      bias[2 * Hd + i] = w.bzd[i];
      bias[3 * Hd + i] = w.brd[i];
    }
  }
  const int* lo = lo_s;

  auto stage3 = [&](const float* a0, const float* a1, const float* a2, int rows) {
    const float* srcs[3] = {a0, a1, a2};
    const int n = rows * Hd;
    for (int k = 0; k < 3; ++k)
      for (int i = threadIdx.x * 4; i < n; i += blockDim.x * 4)
        *reinterpret_cast<float4*>(R + k * FH * Hd + i) = *reinterpret_cast<const float4*>(srcs[k] + i);
  };
  auto input_terms = [&](const float* Wz, const float* Wr, const float* Wh) {
    __syncthreads();  // everyone is done with R / wsl of the previous phase
    stage3(Wz, Wr, Wh, F);
    for (int idx = threadIdx.x * 4; idx < nn * F; idx += blockDim.x * 4) {
      const int v = idx / F, i = idx - v * F;
      *reinterpret_cast<float4*>(wsl + v * F + i) = *reinterpret_cast<const float4*>(xg + (vb + v) * ldx + i);
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 3 * nn * Hd; idx += blockDim.x) {
      const int k = idx / (nn * Hd), r = idx - k * nn * Hd;
      const int v = r / Hd, j = r - v * Hd;
      xt[(k * nn_max + v) * Hd + j] = dot_lds<CF>(R + k * FH * Hd, Hd, j, wsl + v * F, F);
    }
    __syncthreads();
  };
  const float* Uz = R;
  const float* Ur = R + FH * Hd;
  const float* Uh = R + 2 * FH * Hd;
  const float* xz = xt;
  const float* xr = xt + nn_max * Hd;
  const float* xh = xt + 2 * nn_max * Hd;

  // ================= bottom-up (cluster_tree.py:281-316)
  input_terms(w.Wz, w.Wr, w.Wh);
  stage3(w.Uz, w.Ur, w.Uh, Hd);
  {  // leaves (:289-292): no bias on the leaf gates, as the reference
    const int v0 = lo[0], cnt = (lo[1] - lo[0]) * Hd;
    for (int idx = threadIdx.x; idx < cnt; idx += blockDim.x) {
      const int v = v0 + idx / Hd, j = idx % Hd;
      const float z = sigmoidf_(xz[v * Hd + j]);
      const float c = tanhf(xh[v * Hd + j]);
      hl[v * Hd + j] = (1.0f - z) * c;
    }
  }
  __syncthreads();
  for (int l = 1; l < nl; ++l) {
    const int v0 = lo[l], cnt = (lo[l + 1] - lo[l]) * Hd;
    // phase A: ws = sum_c sigmoid(Wr x + Ur h_c + br) * h_c ; hsum kept in hdl (free until top-down)
    for (int idx = threadIdx.x; idx < cnt; idx += blockDim.x) {
      const int v = v0 + idx / Hd, j = idx % Hd;
      const float a = xr[v * Hd + j] + bias[Hd + j];
      float fa = 0.f, hs = 0.f;
      for (int ci = coff_l[v]; ci < coff_l[v + 1]; ++ci) {
        const float* hc = hl + cidx_l[ci] * Hd;
        fa += sigmoidf_(a + dot_lds<CH>(Ur, Hd, j, hc, Hd)) * hc[j];
        hs += hc[j];
      }
      wsl[v * Hd + j] = fa;
      hdl[v * Hd + j] = hs;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < cnt; idx += blockDim.x) {
      const int v = v0 + idx / Hd, j = idx % Hd;
      const float hs = hdl[v * Hd + j];
      const float z = sigmoidf_(xz[v * Hd + j] + bias[j] + dot_lds<CH>(Uz, Hd, j, hdl + v * Hd, Hd));
      const float c = tanhf(xh[v * Hd + j] + dot_lds<CH>(Uh, Hd, j, wsl + v * Hd, Hd));
      hl[v * Hd + j] = z * hs + (1.0f - z) * c;
    }
    __syncthreads();
  }
  for (int idx = threadIdx.x; idx < nn * Hd; idx += blockDim.x)
    hg[(vb + idx / Hd) * ldh + idx % Hd] = hl[idx];

  // ================= top-down (cluster_tree.py:189-200, 318-352)
  input_terms(w.Wzd, w.Wrd, w.Whd);
  stage3(w.Uzd, w.Urd, w.Uhd, Hd);
  {
    const int v0 = lo[nl - 1], cnt = (lo[nl] - lo[nl - 1]) * Hd;
    for (int idx = threadIdx.x; idx < cnt; idx += blockDim.x) hdl[v0 * Hd + idx] = hl[v0 * Hd + idx];  // roots copy h
  }
  __syncthreads();
  for (int l = nl - 2; l >= 0; --l) {
    const int v0 = lo[l], cnt = (lo[l + 1] - lo[l]) * Hd;
    for (int idx = threadIdx.x; idx < cnt; idx += blockDim.x) {
      const int v = v0 + idx / Hd, j = idx % Hd;
      const float* hf = hdl + par_l[v] * Hd;
      const float r = sigmoidf_(xr[v * Hd + j] + dot_lds<CH>(Ur, Hd, j, hf, Hd) + bias[3 * Hd + j]);
      wsl[v * Hd + j] = r * hf[j];
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < cnt; idx += blockDim.x) {
      const int v = v0 + idx / Hd, j = idx % Hd;
      const float* hf = hdl + par_l[v] * Hd;
      const float z = sigmoidf_(xz[v * Hd + j] + dot_lds<CH>(Uz, Hd, j, hf, Hd) + bias[2 * Hd + j]);
      const float c = tanhf(xh[v * Hd + j] + dot_lds<CH>(Uh, Hd, j, wsl + v * Hd, Hd));
      hdl[v * Hd + j] = z * hf[j] + (1.0f - z) * c;
    }
    __syncthreads();
  }
  for (int idx = threadIdx.x; idx < nn * Hd; idx += blockDim.x)
    hdg[(vb + idx / Hd) * ldh + idx % Hd] = hdl[idx];
}


// ---------------------------------------------------------------------------
// Tree-GRU, wide variant (Action Genome head: F = Hd = 256; any Hd in {64,128,256,512,1024}).  One direction's
// matrices (6 x 256 KB) fit neither LDS nor the per-lane dot products of tree_gru_kernel (one L2 round trip per 4
// MACs on a ~40-phase dependent chain: 11 ms at 50 objects per clip).  Here every matrix-vector product of the pass is
// batched over up to TW_NV nodes and the matrix is streamed from L2 exactly once per batch:
//   * tree_xterms_kernel: the input terms W{z,r,h}{,d} x_v of ALL nodes, groups and both directions - no dependence
//     between nodes, one workgroup per (8 nodes, group, matrix), the whole chip;
//   * tree_gru_wide_kernel: one workgroup of 1024 threads per (tree, group) walks the levels.  A product U v is
//     computed for the level's nodes 8 at a time: the vectors sit in LDS (broadcast reads), thread (column j, K-slice
//     s) reads its rows of column j once (coalesced) and keeps 8 partial sums, the K-slices are added in fixed order
//     through LDS.  U_r h_c is computed when h_c is final (once per node instead of once per edge), U_z is applied
//     to the sum of the children states (linear), and the top-down gates take U_{r,z}d hd_f per PARENT.
// Workspace: 6 input-term planes + 2 product planes of n_nodes x G*Hd f32 (trace_tree_gru_workspace_bytes).
// ---------------------------------------------------------------------------
constexpr int TW_THREADS = 1024;
constexpr int TW_NV = 8;

// partial sums over this thread's K-slice rows [i0, i0 + rows) of column j; vs: LDS vectors (n, ldv)
__device__ __forceinline__ void tw_partial(const float* __restrict__ Mt, int Hd, int j, int i0, int rows,
                                           const float* vs, int ldv, int kk, float (&acc)[TW_NV]) {
#pragma unroll
  for (int n = 0; n < TW_NV; ++n) acc[n] = 0.f;
  const float* mp = Mt + (long long)i0 * Hd + j;
#pragma unroll 2
  for (int i = 0; i < rows; i += 4) {
    const float m0 = mp[0], m1 = mp[Hd], m2 = mp[2 * Hd], m3 = mp[3 * Hd];
    mp += 4 * Hd;
#pragma unroll
    for (int n = 0; n < TW_NV; ++n) {
      if (n < kk) {
        const float4 x = *reinterpret_cast<const float4*>(vs + n * ldv + i0 + i);
        acc[n] += m0 * x.x;
        acc[n] += m1 * x.y;
        acc[n] += m2 * x.z;
        acc[n] += m3 * x.w;
      }
    }
  }
}

struct TreeWideParams {
  TreeParams t;
  float* xt;        // 6 planes (z, r, h up; z, r, h down) of n_nodes x G*Hd
  float* pa;        // product plane A: U_r h_c (bottom-up), U_rd hd_f (top-down)
  float* pb;        // product plane B: U_zd hd_f
  int n_nodes;
};

__global__ __launch_bounds__(TW_THREADS) void tree_xterms_kernel(const TreeWideParams q) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const TreeParams& p = q.t;
  const int F = p.F, Hd = p.Hd;
  const int g = blockIdx.y, m = blockIdx.z;
  const int v0 = blockIdx.x * TW_NV;
  const int kk = min(TW_NV, q.n_nodes - v0);
  const TreeW& w = p.w[g];
  const float* Wm = m == 0 ? w.Wz : m == 1 ? w.Wr : m == 2 ? w.Wh : m == 3 ? w.Wzd : m == 4 ? w.Wrd : w.Whd;
  float* vs = sm;                       // [TW_NV][F]
  float* part = sm + TW_NV * F;         // [slices][TW_NV][Hd]
  const long long ldx = (long long)p.G * F, ldh = (long long)p.G * Hd;
  for (int e = threadIdx.x * 4; e < kk * F; e += TW_THREADS * 4) {
    const int n = e / F, i = e - n * F;
    *reinterpret_cast<float4*>(vs + n * F + i) = *reinterpret_cast<const float4*>(p.x + (v0 + n) * ldx + (long long)g * F + i);
  }
  __syncthreads();
  const int n_slices = TW_THREADS / Hd;
  const int j = threadIdx.x % Hd, sl = threadIdx.x / Hd;
  const int rows = F / n_slices;
  float acc[TW_NV];
  tw_partial(Wm, Hd, j, sl * rows, rows, vs, F, kk, acc);
#pragma unroll
  for (int n = 0; n < TW_NV; ++n) part[(sl * TW_NV + n) * Hd + j] = acc[n];
  __syncthreads();
  float* out = q.xt + (long long)m * q.n_nodes * ldh + (long long)g * Hd;
  for (int e = threadIdx.x; e < kk * Hd; e += TW_THREADS) {
    const int n = e / Hd, c = e - n * Hd;
    float s = 0.f;
    for (int k = 0; k < n_slices; ++k) s += part[(k * TW_NV + n) * Hd + c];
    out[(v0 + n) * ldh + c] = s;
  }
}

__global__ __launch_bounds__(TW_THREADS) void tree_gru_wide_kernel(const TreeWideParams q) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const TreeParams& p = q.t;
  const int t = blockIdx.x, g = blockIdx.y;
  const TreeW& w = p.w[g];
  const int Hd = p.Hd;
  const long long ldh = (long long)p.G * Hd;
  const long long plane = (long long)q.n_nodes * ldh;
  float* vsA = sm;                          // [TW_NV][Hd]
  float* vsB = vsA + TW_NV * Hd;            // [TW_NV][Hd]
  float* partA = vsB + TW_NV * Hd;          // [slices][TW_NV][Hd] = TW_NV * 1024
  float* partB = partA + TW_NV * TW_THREADS;
  __shared__ int lo[TRACE_TREE_MAX_LEVELS + 2];
  const int nl = p.n_levels[t];
  for (int i = threadIdx.x; i <= TRACE_TREE_MAX_LEVELS; i += TW_THREADS)
    lo[i] = p.level_off[(long long)t * (TRACE_TREE_MAX_LEVELS + 1) + i];
  float* h = p.h + (long long)g * Hd;
  float* hd = p.hd + (long long)g * Hd;
  float* pa = q.pa + (long long)g * Hd;
  float* pb = q.pb + (long long)g * Hd;
  const float* xz = q.xt + (long long)g * Hd;
  const float* xr = xz + plane;
  const float* xh = xr + plane;
  const int n_slices = TW_THREADS / Hd;
  const int j = threadIdx.x % Hd, sl = threadIdx.x / Hd;
  const int rows = Hd / n_slices, i0 = sl * rows;
  float acc[TW_NV];
  auto put = [&](float* part) {
#pragma unroll
    for (int n = 0; n < TW_NV; ++n) part[(sl * TW_NV + n) * Hd + j] = acc[n];
  };
  auto total = [&](const float* part, int n, int c) {
    float s = 0.f;
    for (int k = 0; k < n_slices; ++k) s += part[(k * TW_NV + n) * Hd + c];
    return s;
  };
  __syncthreads();

  // ================= bottom-up (cluster_tree.py:281-316)
  for (int l = 0; l < nl; ++l) {
    for (int v0 = lo[l]; v0 < lo[l + 1]; v0 += TW_NV) {
      const int kk = min(TW_NV, lo[l + 1] - v0);
      if (l == 0) {
        // leaves (:289-292): no bias on the leaf gates, as the reference
        for (int e = threadIdx.x; e < kk * Hd; e += TW_THREADS) {
          const int n = e / Hd, c = e - n * Hd;
          const long long o = (v0 + n) * ldh + c;
          const float z = sigmoidf_(xz[o]);
          const float hv = (1.0f - z) * tanhf(xh[o]);
          h[o] = hv;
          vsB[n * Hd + c] = hv;
        }
        __syncthreads();
      } else {
        // gated and plain sums over the children (:296-306); U_r h_c was stored when h_c became final
        for (int e = threadIdx.x; e < kk * Hd; e += TW_THREADS) {
          const int n = e / Hd, c = e - n * Hd;
          const int v = v0 + n;
          const float a = xr[v * ldh + c] + w.br[c];
          float fa = 0.f, hs = 0.f;
          for (int ci = p.child_off[v]; ci < p.child_off[v + 1]; ++ci) {
            const long long oc = (long long)p.child_idx[ci] * ldh + c;
            const float hc = h[oc];
            fa += sigmoidf_(a + pa[oc]) * hc;
            hs += hc;
          }
          vsA[n * Hd + c] = hs;
          vsB[n * Hd + c] = fa;
        }
        __syncthreads();
        tw_partial(w.Uz, Hd, j, i0, rows, vsA, Hd, kk, acc);
        put(partA);
        tw_partial(w.Uh, Hd, j, i0, rows, vsB, Hd, kk, acc);
        put(partB);
        __syncthreads();
        for (int e = threadIdx.x; e < kk * Hd; e += TW_THREADS) {
          const int n = e / Hd, c = e - n * Hd;
          const long long o = (v0 + n) * ldh + c;
          const float z = sigmoidf_(xz[o] + w.bz[c] + total(partA, n, c));
          const float cand = tanhf(xh[o] + total(partB, n, c));
          const float hv = z * vsA[n * Hd + c] + (1.0f - z) * cand;
          h[o] = hv;
          vsB[n * Hd + c] = hv;
        }
        __syncthreads();
      }
      if (l < nl - 1) {   // the parent's reset gate needs U_r h_v
        tw_partial(w.Ur, Hd, j, i0, rows, vsB, Hd, kk, acc);
        put(partA);
        __syncthreads();
        for (int e = threadIdx.x; e < kk * Hd; e += TW_THREADS) {
          const int n = e / Hd, c = e - n * Hd;
          pa[(v0 + n) * ldh + c] = total(partA, n, c);
        }
      }
      __syncthreads();
    }
  }

  // ================= top-down (cluster_tree.py:189-200, 318-352)
  const float* xzd = xh + plane;
  const float* xrd = xzd + plane;
  const float* xhd = xrd + plane;
  for (int l = nl - 1; l >= 0; --l) {
    for (int v0 = lo[l]; v0 < lo[l + 1]; v0 += TW_NV) {
      const int kk = min(TW_NV, lo[l + 1] - v0);
      if (l == nl - 1) {
        for (int e = threadIdx.x; e < kk * Hd; e += TW_THREADS) {   // roots copy h
          const int n = e / Hd, c = e - n * Hd;
          const long long o = (v0 + n) * ldh + c;
          const float hv = h[o];
          hd[o] = hv;
          vsA[n * Hd + c] = hv;
        }
        __syncthreads();
      } else {
        for (int e = threadIdx.x; e < kk * Hd; e += TW_THREADS) {
          const int n = e / Hd, c = e - n * Hd;
          const int v = v0 + n;
          const long long of = (long long)p.parent[v] * ldh + c;
          const float r = sigmoidf_(xrd[v * ldh + c] + pa[of] + w.brd[c]);
          vsB[n * Hd + c] = r * hd[of];
        }
        __syncthreads();
        tw_partial(w.Uhd, Hd, j, i0, rows, vsB, Hd, kk, acc);
        put(partA);
        __syncthreads();
        for (int e = threadIdx.x; e < kk * Hd; e += TW_THREADS) {
          const int n = e / Hd, c = e - n * Hd;
          const int v = v0 + n;
          const long long o = v * ldh + c, of = (long long)p.parent[v] * ldh + c;
          const float z = sigmoidf_(xzd[o] + pb[of] + w.bzd[c]);
          const float cand = tanhf(xhd[o] + total(partA, n, c));
          const float hv = z * hd[of] + (1.0f - z) * cand;
          hd[o] = hv;
          vsA[n * Hd + c] = hv;
        }
        __syncthreads();
      }
      if (l > 0) {   // the children's gates need U_rd hd_v and U_zd hd_v
        tw_partial(w.Urd, Hd, j, i0, rows, vsA, Hd, kk, acc);
        put(partA);
        tw_partial(w.Uzd, Hd, j, i0, rows, vsA, Hd, kk, acc);
        put(partB);
        __syncthreads();
        for (int e = threadIdx.x; e < kk * Hd; e += TW_THREADS) {
          const int n = e / Hd, c = e - n * Hd;
          pa[(v0 + n) * ldh + c] = total(partA, n, c);
          pb[(v0 + n) * ldh + c] = total(partB, n, c);
        }
      }
      __syncthreads();
    }
  }
}


// ---------------------------------------------------------------------------
// Triplet ranking: lib/datasets_rel/task_evaluation_vg_and_vrd.py:207-228 on device.
//   per pair: the prd_k best predicates (:207-208), score = sbj * obj * prd (:214-215)
//   per clip: the topk best (pair, predicate) cells (:227 argsort_desc()[:topk])
// Order everywhere: descending score, ties by ascending index (the order of a stable descending sort).
// ---------------------------------------------------------------------------
constexpr int RANK_MAX_PRD = 256;      // predicates per pair held in registers (4 per lane)
constexpr int RANK_MAX_CAND = 16384;   // candidates per clip sorted in LDS (128 KiB)

__device__ __forceinline__ bool rank_before(float sa, int ia, float sb, int ib) {
  return (sa > sb) || (sa == sb && ia < ib);
}

// One wave per pair: kk rounds of wave-wide arg-max over the pair's predicate scores.
__global__ __launch_bounds__(256) void rank_pair_topk_kernel(const float* __restrict__ prd, int n_prd, int ld_prd,
                                                             const float* __restrict__ sbj_sc,
                                                             const float* __restrict__ obj_sc, int P, int kk,
                                                             float* __restrict__ cand_score,
                                                             int32_t* __restrict__ cand_prd) {
  const int lane = threadIdx.x & 63;
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= P) return;
  float v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = lane + 64 * j;
    v[j] = i < n_prd ? prd[(long long)p * ld_prd + i] : -__builtin_inff();
  }
  const float so = sbj_sc[p] * obj_sc[p];  // :214
  for (int slot = 0; slot < kk; ++slot) {
    float bv = v[0];
    int bi = lane;
#pragma unroll
    for (int j = 1; j < 4; ++j)
      if (rank_before(v[j], lane + 64 * j, bv, bi)) {
        bv = v[j];
        bi = lane + 64 * j;
      }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ov = __shfl_xor(bv, off);
      const int oi = __shfl_xor(bi, off);
      if (rank_before(ov, oi, bv, bi)) {
        bv = ov;
        bi = oi;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (bi == lane + 64 * j) v[j] = -__builtin_inff();  // taken
    if (lane == 0) {
      cand_score[(long long)p * kk + slot] = so * bv;  // :215
      cand_prd[(long long)p * kk + slot] = bi;
    }
  }
}

__device__ __forceinline__ void rank_bitonic(float* key, int* idx, int npow2) {
  for (int k = 2; k <= npow2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < npow2; i += blockDim.x) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const float si = key[i], sj = key[ixj];
          const int ii = idx[i], ij = idx[ixj];
          const bool up = ((i & k) == 0);
          const bool swap = up ? rank_before(sj, ij, si, ii) : rank_before(si, ii, sj, ij);
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
}

// Top-k by selection instead of a full sort (prd_k = 20: 7 600 candidates per clip, 100 wanted - the full bitonic
// sort of 8 192 slots was 198 us per step on 8 workgroups).  The k-th best score is found by a 4-pass radix select on
// the order-preserving integer image of the scores (LDS histograms; integer atomics, so the result does not depend
// on their order), the candidates above it plus the lowest-indexed ones equal to it are compacted (index order of
// the ties by a prefix count over contiguous chunks) and only those k are sorted.  Same total order as the sort:
// score descending, candidate index ascending.  blockDim.x == 1024, 1 <= k <= 1024, k < n; on return
// key[0..k) / idx[0..k) hold the k best in rank order.
__device__ __forceinline__ unsigned rank_ord(float f) {
  const unsigned u = __float_as_uint(f + 0.0f);   // -0 -> +0: the two compare equal in rank_before
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ void rank_select_sort(float* key, int* idx, int n, int k, int idx_base) {
  __shared__ int hist[256];
  __shared__ int scan[1024];
  __shared__ int res[2];
  __shared__ float selk[1024];
  __shared__ int seli[1024];
  const int tid = threadIdx.x, lane = tid & 63;
  unsigned prefix = 0, mask = 0;
  int need = k;   // how many of the candidates matching `prefix` are still wanted
  for (int shift = 24; shift >= 0; shift -= 8) {
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 1024) {
      const unsigned u = rank_ord(key[i]);
      if ((u & mask) == prefix) atomicAdd(&hist[(u >> shift) & 255], 1);
    }
    __syncthreads();
    if (tid < 64) {   // lane: bins 255 - 4 lane .. 252 - 4 lane, i.e. descending digit
      const int c0 = hist[255 - 4 * lane], c1 = hist[254 - 4 * lane], c2 = hist[253 - 4 * lane], c3 = hist[252 - 4 * lane];
      const int sum = c0 + c1 + c2 + c3;
      int inc = sum;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(inc, off);
        if (lane >= off) inc += t;
      }
      const int exc = inc - sum;
      if (exc < need && inc >= need) {
        int run = exc, d = 255 - 4 * lane;
        if (run + c0 < need) {
          run += c0; --d;
          if (run + c1 < need) {
            run += c1; --d;
            if (run + c2 < need) { run += c2; --d; }
          }
        }
        res[0] = d;
        res[1] = need - run;
      }
    }
    __syncthreads();
    prefix |= (unsigned)res[0] << shift;
    mask |= 255u << shift;
    need = res[1];
  }
  // prefix = the k-th best key; `need` of the candidates equal to it are taken, lowest index first
  const int per = (n + 1023) >> 10;
  const int i0 = min(n, tid * per), i1 = min(n, i0 + per);
  int gt = 0, eq = 0;
  for (int i = i0; i < i1; ++i) {
    const unsigned u = rank_ord(key[i]);
    gt += u > prefix;
    eq += u == prefix;
  }
  const int own = (gt << 16) | eq;   // gt < k <= 1024, eq <= 16384
  scan[tid] = own;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int t = tid >= off ? scan[tid - off] : 0;
    __syncthreads();
    scan[tid] += t;
    __syncthreads();
  }
  const int gt_total = scan[1023] >> 16;
  int go = (scan[tid] - own) >> 16, eo = (scan[tid] - own) & 0xffff;
  int np2 = 64;
  while (np2 < k) np2 <<= 1;
  for (int i = i0; i < i1; ++i) {
    const float f = key[i] + 0.0f;
    const unsigned u = rank_ord(f);
    if (u > prefix) {
      selk[go] = f;
      seli[go] = idx_base + i;
      ++go;
    } else if (u == prefix) {
      if (eo < need) {
        selk[gt_total + eo] = f;
        seli[gt_total + eo] = idx_base + i;
      }
      ++eo;
    }
  }
  for (int i = k + tid; i < np2; i += 1024) {
    selk[i] = -__builtin_inff();
    seli[i] = 0x7fffffff;
  }
  __syncthreads();
  rank_bitonic(selk, seli, np2);
  for (int i = tid; i < k; i += 1024) {
    key[i] = selk[i];
    idx[i] = seli[i];
  }
  __syncthreads();
}

// One workgroup per clip: bitonic sort of the clip's P_b*kk candidates in LDS, first topk written out.
__global__ __launch_bounds__(1024) void rank_clip_topk_kernel(const float* __restrict__ cand_score,
                                                              const int32_t* __restrict__ cand_prd,
                                                              const int64_t* __restrict__ poff, int kk,
                                                              const int64_t* __restrict__ sbj_lab,
                                                              const int64_t* __restrict__ obj_lab, int topk,
                                                              int npow2, float* __restrict__ top_scores,
                                                              int64_t* __restrict__ top_pair,
                                                              int64_t* __restrict__ top_prd,
                                                              int64_t* __restrict__ top_sbj,
                                                              int64_t* __restrict__ top_obj) {
  extern __shared__ __attribute__((aligned(16))) char rank_smem[];
  float* key = (float*)rank_smem;
  int* idx = (int*)(rank_smem + (size_t)npow2 * sizeof(float));
  const int b = blockIdx.x;
  const long long p0 = poff[b], p1 = poff[b + 1];
  const int n = (int)(p1 - p0) * kk;
  const float* cs = cand_score + p0 * kk;
  for (int i = threadIdx.x; i < npow2; i += blockDim.x) {
    key[i] = i < n ? cs[i] : -__builtin_inff();
    idx[i] = i < n ? i : 0x7fffffff;
  }
  __syncthreads();
  if (n > 1024 && topk <= 1024 && blockDim.x == 1024) rank_select_sort(key, idx, n, topk, 0);   // uniform
  else rank_bitonic(key, idx, npow2);
  for (int j = threadIdx.x; j < topk; j += blockDim.x) {
    const long long o = (long long)b * topk + j;
    if (j < n) {
      const int c = idx[j];
      const long long pair = c / kk;
      top_scores[o] = key[j];
      top_pair[o] = pair;
      top_prd[o] = cand_prd[p0 * kk + c];
      top_sbj[o] = sbj_lab[p0 + pair];
      top_obj[o] = obj_lab[p0 + pair];
    } else {  // fewer candidates than topk: padding the caller slices off
      top_scores[o] = -1.0f;
      top_pair[o] = 0;
      top_prd[o] = 0;
      top_sbj[o] = 0;
      top_obj[o] = 0;
    }
  }
}

// Two-level ranking for a clip with more than RANK_MAX_CAND candidates (configs[4]: 9 900 pairs x prd_k): workgroup
// (s, b) sorts segment s (RANK_MAX_CAND candidates) of clip b and keeps its topk best as (score, clip-local candidate
// index); rank_merge_topk_kernel then sorts the n_seg x topk survivors of a clip.  Both sort on (score descending,
// candidate index ascending) - the total order of the single-pass kernel - so the result is the same list, ties included.
__global__ __launch_bounds__(1024) void rank_seg_topk_kernel(const float* __restrict__ cand_score,
                                                             const int64_t* __restrict__ poff, int kk, int topk,
                                                             int n_seg, float* __restrict__ seg_score,
                                                             int32_t* __restrict__ seg_idx) {
  extern __shared__ __attribute__((aligned(16))) char rank_smem[];
  float* key = (float*)rank_smem;
  int* idx = (int*)(rank_smem + (size_t)RANK_MAX_CAND * sizeof(float));
  const int b = blockIdx.y, sgm = blockIdx.x;
  const long long p0 = poff[b], p1 = poff[b + 1];
  const long long n = (p1 - p0) * kk;
  const long long lo = (long long)sgm * RANK_MAX_CAND;
  const int cnt = (int)(n - lo < 0 ? 0 : (n - lo > RANK_MAX_CAND ? RANK_MAX_CAND : n - lo));
  const float* cs = cand_score + p0 * kk + lo;
  for (int i = threadIdx.x; i < RANK_MAX_CAND; i += blockDim.x) {
    key[i] = i < cnt ? cs[i] : -__builtin_inff();
    idx[i] = i < cnt ? (int)(lo + i) : 0x7fffffff;
  }
  __syncthreads();
  if (cnt > 1024 && topk <= 1024) rank_select_sort(key, idx, cnt, topk, (int)lo);   // cnt is uniform over the workgroup
  else if (cnt > 0) rank_bitonic(key, idx, RANK_MAX_CAND);
  for (int j = threadIdx.x; j < topk; j += blockDim.x) {
    const long long o = ((long long)b * n_seg + sgm) * topk + j;
    seg_score[o] = j < cnt ? key[j] : -__builtin_inff();
    seg_idx[o] = j < cnt ? idx[j] : 0x7fffffff;
  }
}

__global__ __launch_bounds__(1024) void rank_merge_topk_kernel(const float* __restrict__ seg_score,
                                                               const int32_t* __restrict__ seg_idx, int n_in, int npow2,
                                                               const int32_t* __restrict__ cand_prd,
                                                               const int64_t* __restrict__ poff, int kk,
                                                               const int64_t* __restrict__ sbj_lab,
                                                               const int64_t* __restrict__ obj_lab, int topk,
                                                               float* __restrict__ top_scores, int64_t* __restrict__ top_pair,
                                                               int64_t* __restrict__ top_prd, int64_t* __restrict__ top_sbj,
                                                               int64_t* __restrict__ top_obj) {
  extern __shared__ __attribute__((aligned(16))) char rank_smem[];
  float* key = (float*)rank_smem;
  int* idx = (int*)(rank_smem + (size_t)npow2 * sizeof(float));
  const int b = blockIdx.x;
  const long long p0 = poff[b];
  for (int i = threadIdx.x; i < npow2; i += blockDim.x) {
    key[i] = i < n_in ? seg_score[(long long)b * n_in + i] : -__builtin_inff();
    idx[i] = i < n_in ? seg_idx[(long long)b * n_in + i] : 0x7fffffff;
  }
  __syncthreads();
  rank_bitonic(key, idx, npow2);
  for (int j = threadIdx.x; j < topk; j += blockDim.x) {
    const long long o = (long long)b * topk + j;
    const int c = idx[j];
    if (c != 0x7fffffff) {
      const long long pair = c / kk;
      top_scores[o] = key[j];
      top_pair[o] = pair;
      top_prd[o] = cand_prd[p0 * kk + c];
      top_sbj[o] = sbj_lab[p0 + pair];
      top_obj[o] = obj_lab[p0 + pair];
    } else {
      top_scores[o] = -1.0f;
      top_pair[o] = 0;
      top_prd[o] = 0;
      top_sbj[o] = 0;
      top_obj[o] = 0;
    }
  }
}

}  // namespace trace

using namespace trace;

extern "C" int trace_temporal_attention_ex(const float* qs, const float* qo, const float* o, int n, int T, int D,
                                           int K, float inv_sqrt_d, float* out, uint16_t* out_bf16,
                                           trace_stream_t stream) {
  TRACE_REQUIRE(n >= 0 && T > 0 && D > 0 && K > 0, "temporal_attention: bad shape");
  TRACE_REQUIRE(T <= TA_MAX_T, "temporal_attention: T=%d exceeds %d", T, TA_MAX_T);
  if (n == 0) return 0;
  TRACE_REQUIRE(qs && qo && o && (out || out_bf16), "temporal_attention: null pointer");
  hipLaunchKernelGGL(temporal_attention_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, qs, qo, o,
                     T, D, K, inv_sqrt_d, out, out_bf16);
  return check_launch("temporal_attention");
}

extern "C" int trace_temporal_attention(const float* qs, const float* qo, const float* o, int n, int T,
                                        int D, int K, float inv_sqrt_d, float* out,
                                        trace_stream_t stream) {
  TRACE_REQUIRE(n == 0 || out, "temporal_attention: null pointer");
  return trace_temporal_attention_ex(qs, qo, o, n, T, D, K, inv_sqrt_d, out, nullptr, stream);
}

extern "C" int trace_pair_attention_fused(const uint16_t* kmap, const uint16_t* vmap, int B, int H,
                                          int W, int D, const float* rel_rois,
                                          const int32_t* sbj_inds, const int32_t* obj_inds, int P,
                                          const float* q_s, const float* q_o, float spatial_scale,
                                          int pooled, int so_stride, uint16_t* s_map,
                                          uint16_t* o_map, uint16_t* v_map, trace_stream_t stream) {
  TRACE_REQUIRE(so_stride >= D && so_stride % 8 == 0, "pair_attention: so_stride=%d must be >= D and a multiple of 8", so_stride);
  TRACE_REQUIRE(B > 0 && H > 0 && W > 0 && P >= 0 && pooled > 0, "pair_attention: bad shape");
  TRACE_REQUIRE(D == 64 || D == 128 || D == 256 || D == 512, "pair_attention: D=%d must be 64/128/256/512", D);
  TRACE_REQUIRE(pooled * pooled <= PA_MAX_CELLS, "pair_attention: pooled=%d too large", pooled);
  if (P == 0) return 0;
  TRACE_REQUIRE(kmap && vmap && rel_rois && sbj_inds && obj_inds && q_s && q_o && s_map && o_map && v_map,
                "pair_attention: null pointer");
  size_t lds = 2 * PA_MAX_CELLS * sizeof(float) + (size_t)pooled * pooled * D * 2;
  hipLaunchKernelGGL(pair_attention_kernel, dim3(P), dim3(256), lds, (hipStream_t)stream, kmap, vmap, B,
                     H, W, D, rel_rois, sbj_inds, obj_inds, q_s, q_o, spatial_scale, pooled, so_stride, s_map,
                     o_map, v_map);
  return check_launch("pair_attention_fused");
}

extern "C" int trace_pair_infer_fused(const uint16_t* vab, int B, int H, int W, int D, const float* qk, int ld_qk,
                                      const float* rel_rois, const int32_t* sbj_inds, const int32_t* obj_inds,
                                      const int32_t* partner, int P, const float* bias, float spatial_scale, int pooled,
                                      uint16_t* sp, trace_stream_t stream) {
  TRACE_REQUIRE(B > 0 && H > 0 && W > 0 && P >= 0 && pooled > 0, "pair_infer: bad shape");
  TRACE_REQUIRE(D == 64 || D == 128 || D == 256 || D == 512, "pair_infer: D=%d must be 64/128/256/512", D);
  TRACE_REQUIRE(pooled * pooled <= PA_MAX_CELLS, "pair_infer: pooled=%d too large", pooled);
  TRACE_REQUIRE(ld_qk >= B * H * W, "pair_infer: ld_qk=%d smaller than B*H*W", ld_qk);
  if (P == 0) return 0;
  TRACE_REQUIRE(vab && qk && rel_rois && sbj_inds && obj_inds && bias && sp, "pair_infer: null pointer");
  TRACE_REQUIRE((((uintptr_t)vab | (uintptr_t)sp) & 15) == 0, "pair_infer: vab / sp must be 16-byte aligned");
  hipLaunchKernelGGL(pair_infer_kernel, dim3(P), dim3(256), 0, (hipStream_t)stream, vab, B, H, W, D, qk, ld_qk, rel_rois,
                     sbj_inds, obj_inds, partner, bias, spatial_scale, pooled, sp);
  return check_launch("pair_infer_fused");
}

extern "C" int trace_pair_infer_mfma(const uint16_t* vabT, int B, int H, int W, int D, int ldp, const float* vbias,
                                     const float* qk, int ld_qk, const float* rel_rois, const int32_t* sbj_inds,
                                     const int32_t* obj_inds, const int32_t* partner, int P, const int32_t* work, int n_work,
                                     const float* bias, float spatial_scale, int pooled, uint16_t* sp, trace_stream_t stream) {
  TRACE_REQUIRE(B > 0 && H > 0 && W > 0 && P >= 0 && pooled > 0, "pair_infer_mfma: bad shape");
  TRACE_REQUIRE(D % 32 == 0 && D <= 512, "pair_infer_mfma: D=%d must be a multiple of 32, at most 512", D);
  TRACE_REQUIRE(pooled <= 8 && H <= 16 && W <= 16, "pair_infer_mfma: needs pooled <= 8 and a map of at most 16 x 16 (got %d, %d x %d)",
                pooled, H, W);
  TRACE_REQUIRE(ldp == PM_LDP && (H * W + 7) / 8 * 8 + 24 <= PM_LDP, "pair_infer_mfma: pixel pitch must be %d and hold H*W + 31 pixels", PM_LDP);
  TRACE_REQUIRE(ld_qk >= B * H * W, "pair_infer_mfma: ld_qk=%d smaller than B*H*W", ld_qk);
  if (P == 0) return 0;
  TRACE_REQUIRE(vabT && qk && bias && sp && (work || (rel_rois && sbj_inds && obj_inds)), "pair_infer_mfma: null pointer");
  TRACE_REQUIRE((((uintptr_t)vabT | (uintptr_t)sp | (uintptr_t)work) & 15) == 0, "pair_infer_mfma: vabT / sp / work must be 16-byte aligned");
  TRACE_REQUIRE(!work || (n_work >= 0 && n_work <= P), "pair_infer_mfma: bad work list");
  const int nwg = work ? n_work : P;
  if (nwg == 0) return 0;
  hipLaunchKernelGGL(pair_mfma_kernel, dim3(nwg), dim3(PM_THREADS), 0, (hipStream_t)stream, vabT, B, H, W, D, vbias, qk, ld_qk, rel_rois,
                     sbj_inds, obj_inds, partner, work, bias, spatial_scale, pooled, sp);
  return check_launch("pair_infer_mfma");
}

extern "C" int trace_pair_pool(const uint16_t* vab, int B, int H, int W, int D, const float* rel_rois,
                               const int32_t* partner, int P, float spatial_scale, int pooled, uint16_t* out,
                               trace_stream_t stream) {
  TRACE_REQUIRE(B > 0 && H > 0 && W > 0 && P >= 0 && pooled > 0, "pair_pool: bad shape");
  TRACE_REQUIRE(D == 64 || D == 128 || D == 256 || D == 512, "pair_pool: D=%d must be 64/128/256/512", D);
  TRACE_REQUIRE(pooled * pooled <= PA_MAX_CELLS, "pair_pool: pooled=%d too large", pooled);
  if (P == 0) return 0;
  TRACE_REQUIRE(vab && rel_rois && out, "pair_pool: null pointer");
  TRACE_REQUIRE((((uintptr_t)vab | (uintptr_t)out) & 15) == 0, "pair_pool: vab / out must be 16-byte aligned");
  hipLaunchKernelGGL(pair_pool_kernel, dim3(P), dim3(256), 0, (hipStream_t)stream, vab, B, H, W, D, rel_rois, partner,
                     spatial_scale, pooled, out);
  return check_launch("pair_pool");
}

extern "C" int trace_pair_combine(const uint16_t* pooled_maps, int B, int H, int W, int D, const float* qk, int ld_qk,
                                  const float* rel_rois, const int32_t* sbj_inds, const int32_t* obj_inds,
                                  const int32_t* partner, int P, const float* bias, float spatial_scale, int pooled,
                                  uint16_t* sp, trace_stream_t stream) {
  TRACE_REQUIRE(B > 0 && H > 0 && W > 0 && P >= 0 && pooled > 0, "pair_combine: bad shape");
  TRACE_REQUIRE(D == 64 || D == 128 || D == 256 || D == 512, "pair_combine: D=%d must be 64/128/256/512", D);
  TRACE_REQUIRE(pooled * pooled <= PA_MAX_CELLS, "pair_combine: pooled=%d too large", pooled);
  TRACE_REQUIRE(ld_qk >= B * H * W, "pair_combine: ld_qk=%d smaller than B*H*W", ld_qk);
  if (P == 0) return 0;
  TRACE_REQUIRE(pooled_maps && qk && rel_rois && sbj_inds && obj_inds && bias && sp, "pair_combine: null pointer");
  TRACE_REQUIRE((((uintptr_t)pooled_maps | (uintptr_t)sp) & 15) == 0, "pair_combine: maps / sp must be 16-byte aligned");
  hipLaunchKernelGGL(pair_combine_kernel, dim3(P), dim3(256), 0, (hipStream_t)stream, pooled_maps, B, H, W, D, qk, ld_qk,
                     rel_rois, sbj_inds, obj_inds, partner, bias, spatial_scale, pooled, sp);
  return check_launch("pair_combine");
}

static bool tree_wide_ok(int F, int Hd) {
  if (Hd < 64 || Hd > TW_THREADS || TW_THREADS % Hd != 0) return false;
  const int n_slices = TW_THREADS / Hd;
  return F % (4 * n_slices) == 0 && Hd % (4 * n_slices) == 0 && F <= 1024;
}

extern "C" size_t trace_tree_gru_workspace_bytes(int n_nodes, int F, int Hd, int n_groups) {
  if (n_nodes < 0 || F <= 0 || Hd <= 0 || n_groups <= 0) return 0;
  const size_t plane = (size_t)n_nodes * n_groups * Hd * sizeof(float);
  return (tree_wide_ok(F, Hd) ? 8 : 1) * plane + 256;
}

static int tree_gru_impl(const float* x, int n_nodes, int F, int Hd, int n_groups, int n_trees,
                         const int32_t* tree_off, const int32_t* level_off,
                         const int32_t* n_levels, const int32_t* parent,
                         const int32_t* child_off, const int32_t* child_idx,
                         const trace_tree_gru_weights* weights, int max_tree_nodes, float* h,
                         float* hd, float* workspace, size_t workspace_bytes, trace_stream_t stream) {
  TRACE_REQUIRE(n_nodes >= 0 && F > 0 && Hd > 0 && n_groups > 0 && n_trees >= 0, "tree_gru: bad shape");
  TRACE_REQUIRE(n_groups <= TRACE_TREE_MAX_GROUPS, "tree_gru: more than %d groups", TRACE_TREE_MAX_GROUPS);
  if (n_trees == 0 || n_nodes == 0) return 0;
  TRACE_REQUIRE(x && tree_off && level_off && n_levels && parent && child_off && child_idx && weights &&
                    h && hd && workspace,
                "tree_gru: null pointer");
  TreeParams p;
  memset(&p, 0, sizeof(p));
  p.x = x;
  p.F = F;
  p.Hd = Hd;
  p.G = n_groups;
  p.tree_off = tree_off;
  p.level_off = level_off;
  p.n_levels = n_levels;
  p.parent = parent;
  p.child_off = child_off;
  p.child_idx = child_idx;
  p.h = h;
  p.hd = hd;
  p.ws = workspace;
  for (int g = 0; g < n_groups; ++g) {
    const trace_tree_gru_weights& s = weights[g];
    TRACE_REQUIRE(s.Wz && s.Wr && s.Wh && s.Uz && s.Ur && s.Uh && s.bz && s.br && s.Wzd && s.Wrd &&
                      s.Whd && s.Uzd && s.Urd && s.Uhd && s.bzd && s.brd,
                  "tree_gru: null weight pointer in group %d", g);
    p.w[g] = TreeW{s.Wz, s.Wr, s.Wh, s.Uz, s.Ur, s.Uh, s.bz, s.br,
                   s.Wzd, s.Wrd, s.Whd, s.Uzd, s.Urd, s.Uhd, s.bzd, s.brd};
  }
  TRACE_REQUIRE(F % 4 == 0 && Hd % 4 == 0, "tree_gru: F and Hd must be multiples of 4");
  TRACE_REQUIRE((((uintptr_t)x | (uintptr_t)h | (uintptr_t)hd | (uintptr_t)workspace) & 15) == 0,
                "tree_gru: x/h/hd/workspace must be 16-byte aligned");
  // LDS-resident pass when the caller bounds the tree size and one direction fits (see the kernel)
  if (max_tree_nodes > 0) {
    const int FH = F > Hd ? F : Hd;
    const size_t need = ((size_t)3 * FH * Hd + (size_t)max_tree_nodes * (6 * Hd + (F > Hd ? F - Hd : 0)) + 4 * Hd + 3 * (size_t)max_tree_nodes + 4) *
                        sizeof(float);  // xt(3) + h + hd + ws|x, biases, topology
    if (need <= 156 * 1024) {
      TRACE_LDS_OPT_IN((&tree_gru_lds_kernel<0, 0>), 156 * 1024, "tree_gru");
      TRACE_LDS_OPT_IN((&tree_gru_lds_kernel<64, 64>), 156 * 1024, "tree_gru");
      // the pass is a chain of ~4 x levels barrier-separated phases of (nodes of a level) x Hd outputs: more waves per
      // workgroup = fewer rounds per phase (TRACE_TREE_THREADS: A/B measurements)
      static const int tree_threads = [] {
        const char* e = getenv("TRACE_TREE_THREADS");
        const int v = e ? atoi(e) : 512;
        return (v == 256 || v == 512) ? v : 512;   // 1024 would cap the kernel at 128 VGPRs (spills)
      }();
      if (F == 64 && Hd == 64)
        hipLaunchKernelGGL((tree_gru_lds_kernel<64, 64>), dim3(n_trees, n_groups), dim3(tree_threads), need,
                           (hipStream_t)stream, p, max_tree_nodes);
      else
        hipLaunchKernelGGL((tree_gru_lds_kernel<0, 0>), dim3(n_trees, n_groups), dim3(tree_threads), need,
                           (hipStream_t)stream, p, max_tree_nodes);
      return check_launch("tree_gru");
    }
  }
  const size_t lds = (size_t)(3 * F * Hd + 3 * Hd * Hd) * sizeof(float);
  const size_t plane = (size_t)n_nodes * n_groups * Hd;
  static const bool no_wide = getenv("TRACE_TREE_NO_WIDE") != nullptr;
  if (lds > 128 * 1024 && !no_wide && tree_wide_ok(F, Hd) && workspace_bytes >= 8 * plane * sizeof(float)) {
    // matrices too large for LDS: batched products streamed from L2 (see tree_gru_wide_kernel)
    TreeWideParams q;
    q.t = p;
    q.xt = workspace;
    q.pa = workspace + 6 * plane;
    q.pb = workspace + 7 * plane;
    q.n_nodes = n_nodes;
    const size_t lds_x = (size_t)(TW_NV * F + TW_NV * TW_THREADS) * sizeof(float);
    const size_t lds_w = (size_t)(2 * TW_NV * Hd + 2 * TW_NV * TW_THREADS) * sizeof(float);
    TRACE_LDS_OPT_IN(&tree_xterms_kernel, 160 * 1024 - 256, "tree_gru");
    TRACE_LDS_OPT_IN(&tree_gru_wide_kernel, 160 * 1024 - 256, "tree_gru");
    hipLaunchKernelGGL(tree_xterms_kernel, dim3((n_nodes + TW_NV - 1) / TW_NV, n_groups, 6), dim3(TW_THREADS), lds_x,
                       (hipStream_t)stream, q);
    hipLaunchKernelGGL(tree_gru_wide_kernel, dim3(n_trees, n_groups), dim3(TW_THREADS), lds_w, (hipStream_t)stream, q);
    return check_launch("tree_gru");
  }
  if (lds <= 128 * 1024) {
    TRACE_LDS_OPT_IN(&tree_gru_kernel<true>, 128 * 1024, "tree_gru");
    hipLaunchKernelGGL(tree_gru_kernel<true>, dim3(n_trees, n_groups), dim3(256), lds, (hipStream_t)stream, p);
  } else {
    hipLaunchKernelGGL(tree_gru_kernel<false>, dim3(n_trees, n_groups), dim3(256), 0, (hipStream_t)stream, p);
  }
  return check_launch("tree_gru");
}

extern "C" int trace_tree_gru(const float* x, int n_nodes, int F, int Hd, int n_groups, int n_trees,
                              const int32_t* tree_off, const int32_t* level_off,
                              const int32_t* n_levels, const int32_t* parent,
                              const int32_t* child_off, const int32_t* child_idx,
                              const trace_tree_gru_weights* weights, int max_tree_nodes, float* h,
                              float* hd, float* workspace, trace_stream_t stream) {
  return tree_gru_impl(x, n_nodes, F, Hd, n_groups, n_trees, tree_off, level_off, n_levels, parent, child_off, child_idx,
                       weights, max_tree_nodes, h, hd, workspace, (size_t)(n_nodes > 0 ? n_nodes : 0) * n_groups * Hd * sizeof(float),
                       stream);
}

extern "C" int trace_tree_gru_ex(const float* x, int n_nodes, int F, int Hd, int n_groups, int n_trees,
                                 const int32_t* tree_off, const int32_t* level_off,
                                 const int32_t* n_levels, const int32_t* parent,
                                 const int32_t* child_off, const int32_t* child_idx,
                                 const trace_tree_gru_weights* weights, int max_tree_nodes, float* h,
                                 float* hd, float* workspace, size_t workspace_bytes, trace_stream_t stream) {
  TRACE_REQUIRE(n_nodes >= 0 && n_groups > 0 && Hd > 0, "tree_gru: bad shape");
  TRACE_REQUIRE(workspace_bytes >= (size_t)n_nodes * n_groups * Hd * sizeof(float),
                "tree_gru: workspace of %zu bytes is smaller than n_nodes*n_groups*Hd f32", workspace_bytes);
  return tree_gru_impl(x, n_nodes, F, Hd, n_groups, n_trees, tree_off, level_off, n_levels, parent, child_off, child_idx,
                       weights, max_tree_nodes, h, hd, workspace, workspace_bytes, stream);
}

extern "C" size_t trace_rank_triplets_workspace_bytes(int P_total, int prd_k) {
  if (P_total < 0 || prd_k < 0) return 0;
  return (size_t)P_total * prd_k * 8 + 256;
}

static int rank_segments(int max_pairs_per_clip, int kk) {
  const long long cand = (long long)max_pairs_per_clip * kk;
  return cand <= RANK_MAX_CAND ? 1 : (int)((cand + RANK_MAX_CAND - 1) / RANK_MAX_CAND);
}

extern "C" size_t trace_rank_triplets_workspace_bytes_ex(int P_total, int prd_k, int n_clips, int max_pairs_per_clip,
                                                         int topk) {
  if (P_total < 0 || prd_k < 0 || n_clips < 0 || max_pairs_per_clip < 0 || topk < 0) return 0;
  size_t base = (((size_t)P_total * prd_k * 4 + 255) & ~(size_t)255) * 2 + 256;
  const int n_seg = rank_segments(max_pairs_per_clip, prd_k);
  if (n_seg > 1) base += (((size_t)n_clips * n_seg * topk * 4 + 255) & ~(size_t)255) * 2;
  return base;
}

extern "C" int trace_rank_triplets(const float* prd, int P_total, int n_prd, int ld_prd, const float* sbj_scores,
                                   const float* obj_scores, const int64_t* sbj_labels, const int64_t* obj_labels,
                                   const int64_t* pair_off, int n_clips, int max_pairs_per_clip, int prd_k, int topk,
                                   float* top_scores, int64_t* top_pair, int64_t* top_prd, int64_t* top_sbj,
                                   int64_t* top_obj, void* workspace, size_t workspace_bytes, trace_stream_t stream) {
  TRACE_REQUIRE(P_total >= 0 && n_prd > 0 && ld_prd >= n_prd && n_clips >= 0 && prd_k > 0 && topk > 0 &&
                    max_pairs_per_clip >= 0,
                "rank_triplets: bad shape");
  TRACE_REQUIRE(n_prd <= RANK_MAX_PRD, "rank_triplets: %d predicate classes exceed the supported %d", n_prd, RANK_MAX_PRD);
  const int kk = prd_k < n_prd ? prd_k : n_prd;
  const int n_seg = rank_segments(max_pairs_per_clip, kk);
  TRACE_REQUIRE((long long)n_seg * topk <= RANK_MAX_CAND,
                "rank_triplets: %d pairs x %d predicates per clip need %d segments x topk %d > the %d survivors merged on chip",
                max_pairs_per_clip, kk, n_seg, topk, RANK_MAX_CAND);
  if (n_clips == 0) return 0;
  TRACE_REQUIRE(pair_off && top_scores && top_pair && top_prd && top_sbj && top_obj, "rank_triplets: null pointer");
  TRACE_REQUIRE(P_total == 0 || (prd && sbj_scores && obj_scores && sbj_labels && obj_labels && workspace),
                "rank_triplets: null pointer");
  const size_t need = n_seg > 1 ? trace_rank_triplets_workspace_bytes_ex(P_total, kk, n_clips, max_pairs_per_clip, topk)
                                : trace_rank_triplets_workspace_bytes(P_total, kk);
  TRACE_REQUIRE(workspace_bytes >= need, "rank_triplets: workspace too small (%zu < %zu)", workspace_bytes, need);
  TRACE_REQUIRE(((uintptr_t)workspace & 255) == 0, "rank_triplets: workspace must be 256-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const size_t cand_bytes = ((size_t)P_total * kk * 4 + 255) & ~(size_t)255;
  float* cand_score = (float*)workspace;
  int32_t* cand_prd = (int32_t*)((char*)workspace + cand_bytes);
  if (P_total > 0)
    hipLaunchKernelGGL(rank_pair_topk_kernel, dim3(ceil_div(P_total, 4)), dim3(256), 0, s, prd, n_prd, ld_prd, sbj_scores,
                       obj_scores, P_total, kk, cand_score, cand_prd);
  if (n_seg > 1) {
    const size_t seg_bytes = ((size_t)n_clips * n_seg * topk * 4 + 255) & ~(size_t)255;
    float* seg_score = (float*)((char*)workspace + 2 * cand_bytes);
    int32_t* seg_idx = (int32_t*)((char*)workspace + 2 * cand_bytes + seg_bytes);
    TRACE_LDS_OPT_IN(&rank_seg_topk_kernel, RANK_MAX_CAND * 8, "rank_triplets");
    hipLaunchKernelGGL(rank_seg_topk_kernel, dim3(n_seg, n_clips), dim3(1024), (size_t)RANK_MAX_CAND * 8, s, cand_score,
                       pair_off, kk, topk, n_seg, seg_score, seg_idx);
    const int n_in = n_seg * topk;
    int np2 = 64;
    while (np2 < n_in) np2 <<= 1;
    if ((size_t)np2 * 8 > 64 * 1024) {
      TRACE_LDS_OPT_IN(&rank_merge_topk_kernel, RANK_MAX_CAND * 8, "rank_triplets");
    }
    hipLaunchKernelGGL(rank_merge_topk_kernel, dim3(n_clips), dim3(1024), (size_t)np2 * 8, s, seg_score, seg_idx, n_in, np2,
                       cand_prd, pair_off, kk, sbj_labels, obj_labels, topk, top_scores, top_pair, top_prd, top_sbj, top_obj);
    return check_launch("rank_triplets(segmented)");
  }
  int npow2 = 64;
  while (npow2 < max_pairs_per_clip * kk) npow2 <<= 1;
  const size_t lds = (size_t)npow2 * 8;
  if (lds > 64 * 1024) {
    TRACE_LDS_OPT_IN(&rank_clip_topk_kernel, RANK_MAX_CAND * 8, "rank_triplets");
  }
  hipLaunchKernelGGL(rank_clip_topk_kernel, dim3(n_clips), dim3(1024), lds, s, cand_score, cand_prd, pair_off, kk,
                     sbj_labels, obj_labels, topk, npow2, top_scores, top_pair, top_prd, top_sbj, top_obj);
  return check_launch("rank_triplets");
}
