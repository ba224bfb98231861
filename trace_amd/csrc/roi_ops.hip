// ROIAlign / ROIPool kernels for gfx950.
//
// Reference behaviour: Detectron_pytorch/lib/model/csrc/cuda/ROIAlign_cuda.cu:16-122
// (forward), :125-254 (backward), cpu/ROIAlign_cpu.cpp:17-219 (arithmetic order
// of the oracle), cuda/ROIPool_cuda.cu:16-108.
//
// This file is compiled with -ffp-contract=off: the f32 NCHW forward reproduces
// the CPU reference's rounding sequence bit for bit.
#include "common.h"
#include <stdlib.h>

namespace trace {

struct RoiGeom {
  int batch;
  float start_w, start_h, bin_w, bin_h;
  int grid_h, grid_w;
  float count;
};

// ROIAlign_cuda.cu:78-104 / ROIAlign_cpu.cpp:138-169
__device__ __forceinline__ RoiGeom roi_geom(const float* r, float scale, int PH, int PW, int sr, int B) {
  RoiGeom g;
  // a batch index outside [0,B) is clamped rather than faulting the GPU
  g.batch = min(max((int)r[0], 0), B - 1);
  g.start_w = r[1] * scale;
  g.start_h = r[2] * scale;
  float end_w = r[3] * scale;
  float end_h = r[4] * scale;
  float roi_w = fmaxf(end_w - g.start_w, 1.0f);
  float roi_h = fmaxf(end_h - g.start_h, 1.0f);
  g.bin_h = roi_h / (float)PH;
  g.bin_w = roi_w / (float)PW;
  g.grid_h = (sr > 0) ? sr : (int)ceilf(roi_h / (float)PH);
  g.grid_w = (sr > 0) ? sr : (int)ceilf(roi_w / (float)PW);
  g.count = (float)(g.grid_h * g.grid_w);
  return g;
}

struct Bilin {
  int p1, p2, p3, p4;  // offsets inside one H*W plane (in pixels)
  float w1, w2, w3, w4;
  bool valid;
};

// ROIAlign_cpu.cpp:44-104 / ROIAlign_cuda.cu:22-57
__device__ __forceinline__ Bilin bilin_setup(float y, float x, int H, int W) {
  Bilin b;
  if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) {
    b.p1 = b.p2 = b.p3 = b.p4 = 0;
    b.w1 = b.w2 = b.w3 = b.w4 = 0.f;
    b.valid = false;
    return b;
  }
  if (y <= 0.f) y = 0.f;
  if (x <= 0.f) x = 0.f;
  int y_low = (int)y, x_low = (int)x, y_high, x_high;
  if (y_low >= H - 1) {
    y_high = y_low = H - 1;
    y = (float)y_low;
  } else {
    y_high = y_low + 1;
  }
  if (x_low >= W - 1) {
    x_high = x_low = W - 1;
    x = (float)x_low;
  } else {
    x_high = x_low + 1;
  }
  float ly = y - (float)y_low, lx = x - (float)x_low;
  float hy = 1.0f - ly, hx = 1.0f - lx;
  b.w1 = hy * hx;
  b.w2 = hy * lx;
  b.w3 = ly * hx;
  b.w4 = ly * lx;
  b.p1 = y_low * W + x_low;
  b.p2 = y_low * W + x_high;
  b.p3 = y_high * W + x_low;
  b.p4 = y_high * W + x_high;
  b.valid = true;
  return b;
}

__device__ __forceinline__ float sample_coord(float start, int p, float bin, int i, int grid) {
  // roi_start + p * bin + (i + .5f) * bin / grid   (ROIAlign_cpu.cpp:36-42)
  return start + (float)p * bin + ((float)i + .5f) * bin / (float)grid;
}

// ---------------------------------------------------------------------------
// Forward, NCHW f32 (drop-in _C.roi_align_forward semantics).
// One thread per output element; consecutive lanes write consecutive addresses.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void roi_align_fwd_f32_kernel(
    long long nthreads, const float* __restrict__ in, float scale, int B, int C, int H, int W,
    int PH, int PW, int sr, const float* __restrict__ rois, float* __restrict__ out) {
  for (long long index = (long long)blockIdx.x * blockDim.x + threadIdx.x; index < nthreads;
       index += (long long)blockDim.x * gridDim.x) {
    int pw = (int)(index % PW);
    int ph = (int)((index / PW) % PH);
    int c = (int)((index / PW / PH) % C);
    int n = (int)(index / PW / PH / C);
    RoiGeom g = roi_geom(rois + (long long)n * 5, scale, PH, PW, sr, B);
    const float* plane = in + ((long long)g.batch * C + c) * H * W;
    float acc = 0.f;
    for (int iy = 0; iy < g.grid_h; ++iy) {
      float y = sample_coord(g.start_h, ph, g.bin_h, iy, g.grid_h);
      for (int ix = 0; ix < g.grid_w; ++ix) {
        float x = sample_coord(g.start_w, pw, g.bin_w, ix, g.grid_w);
        Bilin b = bilin_setup(y, x, H, W);
        // ROIAlign_cpu.cpp:204-207 — left-to-right sum, no contraction
        float v = b.w1 * plane[b.p1] + b.w2 * plane[b.p2] + b.w3 * plane[b.p3] + b.w4 * plane[b.p4];
        acc += v;
      }
    }
    out[index] = acc / g.count;
  }
}

// ---------------------------------------------------------------------------
// Backward, NCHW f32 (ROIAlign_cuda.cu:177-254).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void roi_align_bwd_f32_kernel(
    long long nthreads, const float* __restrict__ grad, float scale, int B, int C, int H, int W,
    int PH, int PW, int sr, float* __restrict__ gin, const float* __restrict__ rois) {
  for (long long index = (long long)blockIdx.x * blockDim.x + threadIdx.x; index < nthreads;
       index += (long long)blockDim.x * gridDim.x) {
    int pw = (int)(index % PW);
    int ph = (int)((index / PW) % PH);
    int c = (int)((index / PW / PH) % C);
    int n = (int)(index / PW / PH / C);
    RoiGeom g = roi_geom(rois + (long long)n * 5, scale, PH, PW, sr, B);
    float* plane = gin + ((long long)g.batch * C + c) * H * W;
    float top = grad[index];
    for (int iy = 0; iy < g.grid_h; ++iy) {
      float y = sample_coord(g.start_h, ph, g.bin_h, iy, g.grid_h);
      for (int ix = 0; ix < g.grid_w; ++ix) {
        float x = sample_coord(g.start_w, pw, g.bin_w, ix, g.grid_w);
        Bilin b = bilin_setup(y, x, H, W);
        if (b.valid) {
          atomicAdd(plane + b.p1, top * b.w1 / g.count);
          atomicAdd(plane + b.p2, top * b.w2 / g.count);
          atomicAdd(plane + b.p3, top * b.w3 / g.count);
          atomicAdd(plane + b.p4, top * b.w4 / g.count);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// ROIPool forward/backward (ROIPool_cuda.cu:16-108).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void roi_pool_fwd_f32_kernel(
    long long nthreads, const float* __restrict__ in, float scale, int B, int C, int H, int W,
    int PH, int PW, const float* __restrict__ rois, float* __restrict__ out,
    int32_t* __restrict__ argmax) {
  for (long long index = (long long)blockIdx.x * blockDim.x + threadIdx.x; index < nthreads;
       index += (long long)blockDim.x * gridDim.x) {
    int pw = (int)(index % PW);
    int ph = (int)((index / PW) % PH);
    int c = (int)((index / PW / PH) % C);
    int n = (int)(index / PW / PH / C);
    const float* r = rois + (long long)n * 5;
    int b = min(max((int)r[0], 0), B - 1);
    int rsw = (int)roundf(r[1] * scale);
    int rsh = (int)roundf(r[2] * scale);
    int rew = (int)roundf(r[3] * scale);
    int reh = (int)roundf(r[4] * scale);
    int rw = max(rew - rsw + 1, 1);
    int rh = max(reh - rsh + 1, 1);
    float bin_h = (float)rh / (float)PH;
    float bin_w = (float)rw / (float)PW;
    int hstart = (int)floorf((float)ph * bin_h);
    int wstart = (int)floorf((float)pw * bin_w);
    int hend = (int)ceilf((float)(ph + 1) * bin_h);
    int wend = (int)ceilf((float)(pw + 1) * bin_w);
    hstart = min(max(hstart + rsh, 0), H);
    hend = min(max(hend + rsh, 0), H);
    wstart = min(max(wstart + rsw, 0), W);
    wend = min(max(wend + rsw, 0), W);
    bool is_empty = (hend <= hstart) || (wend <= wstart);
    float maxval = is_empty ? 0.f : -3.402823466e+38f;
    int maxidx = -1;
    const float* plane = in + ((long long)b * C + c) * H * W;
    for (int h = hstart; h < hend; ++h)
      for (int w = wstart; w < wend; ++w) {
        int bi = h * W + w;
        float v = plane[bi];
        if (v > maxval) {
          maxval = v;
          maxidx = bi;
        }
      }
    out[index] = maxval;
    argmax[index] = maxidx;
  }
}

__global__ __launch_bounds__(256) void roi_pool_bwd_f32_kernel(
    long long nthreads, const float* __restrict__ grad, const int32_t* __restrict__ argmax, int B,
    int C, int H, int W, int PH, int PW, float* __restrict__ gin, const float* __restrict__ rois) {
  for (long long index = (long long)blockIdx.x * blockDim.x + threadIdx.x; index < nthreads;
       index += (long long)blockDim.x * gridDim.x) {
    int c = (int)((index / PW / PH) % C);
    int n = (int)(index / PW / PH / C);
    int b = min(max((int)rois[(long long)n * 5], 0), B - 1);
    int am = argmax[index];
    if (am != -1) atomicAdd(gin + ((long long)b * C + c) * H * W + am, grad[index]);
  }
}

// ---------------------------------------------------------------------------
// The same four NCHW ops in float64: the reference's extension dispatches on float / double
// (AT_DISPATCH_FLOATING_TYPES, ROIAlign_cuda.cu:283,330; ROIPool_cuda.cu:129,178; ROIAlign_cpu.cpp:243), rois in the
// input's dtype.  Kept apart from the float32 kernels above (whose bit-exact contract then cannot move): the same
// statements with double in place of float, one thread per output element.
// ---------------------------------------------------------------------------
struct RoiGeomD {
  int batch;
  double start_w, start_h, bin_w, bin_h;
  int grid_h, grid_w;
  double count;
};
__device__ __forceinline__ RoiGeomD roi_geom_d(const double* r, double scale, int PH, int PW, int sr, int B) {
  RoiGeomD g;
  g.batch = min(max((int)r[0], 0), B - 1);
  g.start_w = r[1] * scale;
  g.start_h = r[2] * scale;
  const double end_w = r[3] * scale, end_h = r[4] * scale;
  const double roi_w = fmax(end_w - g.start_w, 1.0), roi_h = fmax(end_h - g.start_h, 1.0);
  g.bin_h = roi_h / (double)PH;
  g.bin_w = roi_w / (double)PW;
  g.grid_h = (sr > 0) ? sr : (int)ceil(roi_h / (double)PH);
  g.grid_w = (sr > 0) ? sr : (int)ceil(roi_w / (double)PW);
  g.count = (double)(g.grid_h * g.grid_w);
  return g;
}
struct BilinD {
  int p1, p2, p3, p4;
  double w1, w2, w3, w4;
  bool valid;
};
__device__ __forceinline__ BilinD bilin_setup_d(double y, double x, int H, int W) {
  BilinD b;
  if (y < -1.0 || y > (double)H || x < -1.0 || x > (double)W) {
    b.p1 = b.p2 = b.p3 = b.p4 = 0;
    b.w1 = b.w2 = b.w3 = b.w4 = 0.0;
    b.valid = false;
    return b;
  }
  if (y <= 0.0) y = 0.0;
  if (x <= 0.0) x = 0.0;
  int y_low = (int)y, x_low = (int)x, y_high, x_high;
  if (y_low >= H - 1) {
    y_high = y_low = H - 1;
    y = (double)y_low;
  } else {
    y_high = y_low + 1;
  }
  if (x_low >= W - 1) {
    x_high = x_low = W - 1;
    x = (double)x_low;
  } else {
    x_high = x_low + 1;
  }
  const double ly = y - (double)y_low, lx = x - (double)x_low;
  const double hy = 1.0 - ly, hx = 1.0 - lx;
  b.w1 = hy * hx;
  b.w2 = hy * lx;
  b.w3 = ly * hx;
  b.w4 = ly * lx;
  b.p1 = y_low * W + x_low;
  b.p2 = y_low * W + x_high;
  b.p3 = y_high * W + x_low;
  b.p4 = y_high * W + x_high;
  b.valid = true;
  return b;
}
__device__ __forceinline__ double sample_coord_d(double start, int p, double bin, int i, int grid) {
  return start + (double)p * bin + (double)((float)i + .5f) * bin / (double)grid;   // static_cast<T>(iy + .5f)
}

__global__ __launch_bounds__(256) void roi_align_fwd_f64_kernel(
    long long nthreads, const double* __restrict__ in, double scale, int B, int C, int H, int W,
    int PH, int PW, int sr, const double* __restrict__ rois, double* __restrict__ out) {
  for (long long index = (long long)blockIdx.x * blockDim.x + threadIdx.x; index < nthreads;
       index += (long long)blockDim.x * gridDim.x) {
    const int pw = (int)(index % PW), ph = (int)((index / PW) % PH);
    const int c = (int)((index / PW / PH) % C), n = (int)(index / PW / PH / C);
    const RoiGeomD g = roi_geom_d(rois + (long long)n * 5, scale, PH, PW, sr, B);
    const double* plane = in + ((long long)g.batch * C + c) * H * W;
    double acc = 0.0;
    for (int iy = 0; iy < g.grid_h; ++iy) {
      const double y = sample_coord_d(g.start_h, ph, g.bin_h, iy, g.grid_h);
      for (int ix = 0; ix < g.grid_w; ++ix) {
        const double x = sample_coord_d(g.start_w, pw, g.bin_w, ix, g.grid_w);
        const BilinD b = bilin_setup_d(y, x, H, W);
        acc += b.w1 * plane[b.p1] + b.w2 * plane[b.p2] + b.w3 * plane[b.p3] + b.w4 * plane[b.p4];
      }
    }
    out[index] = acc / g.count;
  }
}

__global__ __launch_bounds__(256) void roi_align_bwd_f64_kernel(
    long long nthreads, const double* __restrict__ grad, double scale, int B, int C, int H, int W,
    int PH, int PW, int sr, double* __restrict__ gin, const double* __restrict__ rois) {
  for (long long index = (long long)blockIdx.x * blockDim.x + threadIdx.x; index < nthreads;
       index += (long long)blockDim.x * gridDim.x) {
    const int pw = (int)(index % PW), ph = (int)((index / PW) % PH);
    const int c = (int)((index / PW / PH) % C), n = (int)(index / PW / PH / C);
    const RoiGeomD g = roi_geom_d(rois + (long long)n * 5, scale, PH, PW, sr, B);
    double* plane = gin + ((long long)g.batch * C + c) * H * W;
    const double top = grad[index];
    for (int iy = 0; iy < g.grid_h; ++iy) {
      const double y = sample_coord_d(g.start_h, ph, g.bin_h, iy, g.grid_h);
      for (int ix = 0; ix < g.grid_w; ++ix) {
        const double x = sample_coord_d(g.start_w, pw, g.bin_w, ix, g.grid_w);
        const BilinD b = bilin_setup_d(y, x, H, W);
        if (b.valid) {
          atomicAdd(plane + b.p1, top * b.w1 / g.count);
          atomicAdd(plane + b.p2, top * b.w2 / g.count);
          atomicAdd(plane + b.p3, top * b.w3 / g.count);
          atomicAdd(plane + b.p4, top * b.w4 / g.count);
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void roi_pool_fwd_f64_kernel(
    long long nthreads, const double* __restrict__ in, double scale, int B, int C, int H, int W,
    int PH, int PW, const double* __restrict__ rois, double* __restrict__ out, int32_t* __restrict__ argmax) {
  for (long long index = (long long)blockIdx.x * blockDim.x + threadIdx.x; index < nthreads;
       index += (long long)blockDim.x * gridDim.x) {
    const int pw = (int)(index % PW), ph = (int)((index / PW) % PH);
    const int c = (int)((index / PW / PH) % C), n = (int)(index / PW / PH / C);
    const double* r = rois + (long long)n * 5;
    const int b = min(max((int)r[0], 0), B - 1);
    const int rsw = (int)round(r[1] * scale), rsh = (int)round(r[2] * scale);
    const int rew = (int)round(r[3] * scale), reh = (int)round(r[4] * scale);
    const int rw = max(rew - rsw + 1, 1), rh = max(reh - rsh + 1, 1);
    const double bin_h = (double)rh / (double)PH, bin_w = (double)rw / (double)PW;
    int hstart = (int)floor((double)ph * bin_h), wstart = (int)floor((double)pw * bin_w);
    int hend = (int)ceil((double)(ph + 1) * bin_h), wend = (int)ceil((double)(pw + 1) * bin_w);
    hstart = min(max(hstart + rsh, 0), H);
    hend = min(max(hend + rsh, 0), H);
    wstart = min(max(wstart + rsw, 0), W);
    wend = min(max(wend + rsw, 0), W);
    const bool is_empty = (hend <= hstart) || (wend <= wstart);
    double maxval = is_empty ? 0.0 : -3.402823466e+38;   // -FLT_MAX for every dtype (ROIPool_cuda.cu:59)
    int maxidx = -1;
    const double* plane = in + ((long long)b * C + c) * H * W;
    for (int h = hstart; h < hend; ++h)
      for (int w = wstart; w < wend; ++w) {
        const int bi = h * W + w;
        const double v = plane[bi];
        if (v > maxval) {
          maxval = v;
          maxidx = bi;
        }
      }
    out[index] = maxval;
    argmax[index] = maxidx;
  }
}

__global__ __launch_bounds__(256) void roi_pool_bwd_f64_kernel(
    long long nthreads, const double* __restrict__ grad, const int32_t* __restrict__ argmax, int B,
    int C, int H, int W, int PH, int PW, double* __restrict__ gin, const double* __restrict__ rois) {
  for (long long index = (long long)blockIdx.x * blockDim.x + threadIdx.x; index < nthreads;
       index += (long long)blockDim.x * gridDim.x) {
    const int c = (int)((index / PW / PH) % C), n = (int)(index / PW / PH / C);
    const int b = min(max((int)rois[(long long)n * 5], 0), B - 1);
    const int am = argmax[index];
    if (am != -1) atomicAdd(gin + ((long long)b * C + c) * H * W + am, grad[index]);
  }
}

// ---------------------------------------------------------------------------
// Forward, channels-last bf16: input (B,H,W,C), out (K,PH*PW,C).
// One workgroup per (roi, bin); lane owns 8 consecutive channels (16 B) so each
// corner read and the output write are 1 KiB-per-wave coalesced accesses; the
// sample geometry depends only on blockIdx (wave-uniform).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void roi_align_fwd_nhwc_bf16_kernel(
    const uint16_t* __restrict__ in, int B, int H, int W, int C, const float* __restrict__ rois,
    float scale, int PH, int PW, int sr, uint16_t* __restrict__ out) {
  const int bin = blockIdx.x;  // ph*PW+pw
  const int n = blockIdx.y;
  const int ph = bin / PW, pw = bin - ph * PW;
  RoiGeom g = roi_geom(rois + (long long)n * 5, scale, PH, PW, sr, B);
  const uint16_t* img = in + (long long)g.batch * H * W * C;
  uint16_t* dst = out + ((long long)n * PH * PW + bin) * C;
  const float inv_count = 1.0f / g.count;
  for (int c0 = threadIdx.x * 8; c0 < C; c0 += blockDim.x * 8) {
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int iy = 0; iy < g.grid_h; ++iy) {
      float y = sample_coord(g.start_h, ph, g.bin_h, iy, g.grid_h);
      for (int ix = 0; ix < g.grid_w; ++ix) {
        float x = sample_coord(g.start_w, pw, g.bin_w, ix, g.grid_w);
        Bilin b = bilin_setup(y, x, H, W);
        if (!b.valid) continue;
        uint4 v1 = *reinterpret_cast<const uint4*>(img + (long long)b.p1 * C + c0);
        uint4 v2 = *reinterpret_cast<const uint4*>(img + (long long)b.p2 * C + c0);
        uint4 v3 = *reinterpret_cast<const uint4*>(img + (long long)b.p3 * C + c0);
        uint4 v4 = *reinterpret_cast<const uint4*>(img + (long long)b.p4 * C + c0);
        const uint32_t* a1 = &v1.x;
        const uint32_t* a2 = &v2.x;
        const uint32_t* a3 = &v3.x;
        const uint32_t* a4 = &v4.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float l = b.w1 * __uint_as_float(a1[j] << 16) + b.w2 * __uint_as_float(a2[j] << 16) +
                    b.w3 * __uint_as_float(a3[j] << 16) + b.w4 * __uint_as_float(a4[j] << 16);
          float h = b.w1 * __uint_as_float(a1[j] & 0xffff0000u) +
                    b.w2 * __uint_as_float(a2[j] & 0xffff0000u) +
                    b.w3 * __uint_as_float(a3[j] & 0xffff0000u) +
                    b.w4 * __uint_as_float(a4[j] & 0xffff0000u);
          acc[2 * j] += l;
          acc[2 * j + 1] += h;
        }
      }
    }
    uint4 o;
    o.x = pack_bf16x2(acc[0] * inv_count, acc[1] * inv_count);
    o.y = pack_bf16x2(acc[2] * inv_count, acc[3] * inv_count);
    o.z = pack_bf16x2(acc[4] * inv_count, acc[5] * inv_count);
    o.w = pack_bf16x2(acc[6] * inv_count, acc[7] * inv_count);
    *reinterpret_cast<uint4*>(dst + c0) = o;
  }
}

// ---------------------------------------------------------------------------
// ROIAlign + AdaptiveAvgPool2d((1,1)) fused (Tail_Layer_3D.map2vec with I3D_DC5,
// lib/i3d/box_head3d.py:59-61): out (K,C) bf16 = mean over the PH*PW bins of the
// ROIAlign output, without writing the (K,PH,PW,C) tensor.  One workgroup per roi.
//
// The mean over all bins of all bilinear samples is a weighted sum of map pixels, and because the
// sample grid is a Cartesian product (every y sample meets every x sample, and a sample is dropped
// iff its y OR its x is out of range) the pixel weight factorises: W[py][px] = Wy[py] * Wx[px].
// Wy / Wx are built once per roi in LDS (a few dozen 1-D samples each, fixed summation order, no
// atomics), then every lane accumulates its 8 channels over the roi's pixel box: (roi_h+2)*(roi_w+2)
// 16-byte loads instead of 4*PH*PW*grid_h*grid_w.
// ---------------------------------------------------------------------------
constexpr int RA_MAX_DIM = 1024;  // map height / width limit of the separable kernel

struct Lin1 {
  int lo, hi;
  float wl, wh;
  bool valid;
};
// 1-D half of bilinear_interpolate (ROIAlign_cuda.cu:22-57): the y rules and the x rules are identical
__device__ __forceinline__ Lin1 lin_setup(float v, int n) {
  Lin1 r;
  r.valid = !(v < -1.0f || v > (float)n);
  if (v <= 0.f) v = 0.f;
  r.lo = (int)v;
  if (r.lo >= n - 1) {
    r.hi = r.lo = n - 1;
    v = (float)r.lo;
  } else {
    r.hi = r.lo + 1;
  }
  r.wh = v - (float)r.lo;
  r.wl = 1.0f - r.wh;
  return r;
}

__global__ __launch_bounds__(256) void roi_align_avg_nhwc_bf16_kernel(
    const uint16_t* __restrict__ in, int B, int H, int W, int C, const float* __restrict__ rois,
    float scale, int PH, int PW, int sr, uint16_t* __restrict__ out) {
  __shared__ float wy[RA_MAX_DIM];
  __shared__ float wx[RA_MAX_DIM];
  __shared__ int box[4];
  const int n = blockIdx.x;
  RoiGeom g = roi_geom(rois + (long long)n * 5, scale, PH, PW, sr, B);
  const uint16_t* img = in + (long long)g.batch * H * W * C;
  // pixel p of an axis collects, over that axis' samples, wl where lo == p and wh where hi == p
  for (int p = threadIdx.x; p < H + W; p += blockDim.x) {
    const bool is_y = p < H;
    const int q = is_y ? p : p - H;
    const int bins = is_y ? PH : PW, grid = is_y ? g.grid_h : g.grid_w, dim = is_y ? H : W;
    const float start = is_y ? g.start_h : g.start_w, bsz = is_y ? g.bin_h : g.bin_w;
    float acc = 0.f;
    for (int b = 0; b < bins; ++b)
      for (int i = 0; i < grid; ++i) {
        const Lin1 l = lin_setup(sample_coord(start, b, bsz, i, grid), dim);
        if (!l.valid) continue;
        if (l.lo == q) acc += l.wl;
        if (l.hi == q) acc += l.wh;
      }
    (is_y ? wy : wx)[q] = acc;
  }
  if (threadIdx.x < 4) {
    // pixel range touched by the first / last sample of each axis (samples are monotonic)
    const bool is_y = threadIdx.x < 2;
    const int bins = is_y ? PH : PW, grid = is_y ? g.grid_h : g.grid_w, dim = is_y ? H : W;
    const float start = is_y ? g.start_h : g.start_w, bsz = is_y ? g.bin_h : g.bin_w;
    const bool first = (threadIdx.x & 1) == 0;
    const Lin1 l = lin_setup(first ? sample_coord(start, 0, bsz, 0, grid) : sample_coord(start, bins - 1, bsz, grid - 1, grid), dim);
    box[threadIdx.x] = first ? l.lo : l.hi;
  }
  __syncthreads();
  const int y0 = box[0], y1 = box[1], x0 = box[2], x1 = box[3];
  const float norm = 1.0f / (g.count * (float)(PH * PW));
  for (int c0 = threadIdx.x * 8; c0 < C; c0 += blockDim.x * 8) {
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int py = y0; py <= y1; ++py) {
      const float wyv = wy[py];
      if (wyv == 0.f) continue;
      const uint16_t* row = img + ((long long)py * W) * C + c0;
      float racc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) racc[j] = 0.f;
#pragma unroll 4
      for (int px = x0; px <= x1; ++px) {
        const float w = wx[px];
        const uint4 v = *reinterpret_cast<const uint4*>(row + (long long)px * C);
        const uint32_t* a = &v.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          racc[2 * j] += w * __uint_as_float(a[j] << 16);
          racc[2 * j + 1] += w * __uint_as_float(a[j] & 0xffff0000u);
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += wyv * racc[j];
    }
    uint4 o;
    o.x = pack_bf16x2(acc[0] * norm, acc[1] * norm);
    o.y = pack_bf16x2(acc[2] * norm, acc[3] * norm);
    o.z = pack_bf16x2(acc[4] * norm, acc[5] * norm);
    o.w = pack_bf16x2(acc[6] * norm, acc[7] * norm);
    *reinterpret_cast<uint4*>(out + (long long)n * C + c0) = o;
  }
}

// ---------------------------------------------------------------------------
// Forward, channels-last bf16, 7x7 bins, SEPARABLE form.  The kernel above reads 4 corners per sample per bin
// (49 x grid^2 x 4 sixteen-byte loads per lane and roi) and is L2-bandwidth bound: 514 MB of L2 reads for the
// 38 MB the step's object rois need.  A bin is a Cartesian product of y samples and x samples and a sample is
// dropped iff its y OR its x is out of range (ROIAlign_cuda.cu:22-31), so
//   out[ph][pw] = sum_py sum_px  Wy[ph][py] Wx[pw][px] map[py][px]
// with Wy / Wx the per-axis sums of the 1-D interpolation weights (/ grid).  One wave owns 128 channels of one
// roi (2 per lane, 49 x 2 accumulators): every map pixel of the roi's box is read ONCE, folded into the 7 column
// bins of its row (Wx from LDS, broadcast reads), and a finished row is folded into the row bins with non-zero
// Wy.  (box_h x box_w) loads instead of 196 x grid^2; f32 accumulation in a different order than the kernel
// above (feature-path tolerance; the bit-exact contract is the f32 NCHW kernel's).
// ---------------------------------------------------------------------------
constexpr int RS_MAX_DIM = 64;   // map height / width limit of the separable forward kernel
__global__ __launch_bounds__(64) void roi_align_fwd_sep7_nhwc_bf16_kernel(
    const uint16_t* __restrict__ in, int B, int H, int W, int C, const float* __restrict__ rois,
    float scale, int sr, uint16_t* __restrict__ out) {
  constexpr int P = 7;
  __shared__ __attribute__((aligned(16))) float wyT[RS_MAX_DIM][8];   // [pixel row][bin], bin 7 = padding
  __shared__ __attribute__((aligned(16))) float wxT[RS_MAX_DIM + 16][8];   // 16 zero rows behind the map's last column
  __shared__ int box[4];
  const int n = blockIdx.y;
  const int c0 = blockIdx.x * 128 + threadIdx.x * 2;
  for (int e = threadIdx.x; e < 16 * 8; e += 64) wxT[W + (e >> 3)][e & 7] = 0.f;
  RoiGeom g = roi_geom(rois + (long long)n * 5, scale, P, P, sr, B);
  const uint16_t* img = in + (long long)g.batch * H * W * C + c0;
  for (int e = threadIdx.x; e < (H + W) * 8; e += 64) {
    const int q = e >> 3, b = e & 7;
    const bool is_y = q < H;
    const int pix = is_y ? q : q - H;
    const int grid = is_y ? g.grid_h : g.grid_w, dim = is_y ? H : W;
    const float start = is_y ? g.start_h : g.start_w, bsz = is_y ? g.bin_h : g.bin_w;
    float acc = 0.f;
    if (b < P) {
      for (int i = 0; i < grid; ++i) {
        const Lin1 l = lin_setup(sample_coord(start, b, bsz, i, grid), dim);
        if (!l.valid) continue;
        if (l.lo == pix) acc += l.wl;
        if (l.hi == pix) acc += l.wh;
      }
      acc /= (float)grid;
    }
    (is_y ? wyT : wxT)[pix][b] = acc;
  }
  if (threadIdx.x < 4) {
    // pixel range touched by the first / last sample of each axis (samples are monotonic)
    const bool is_y = threadIdx.x < 2;
    const int grid = is_y ? g.grid_h : g.grid_w, dim = is_y ? H : W;
    const float start = is_y ? g.start_h : g.start_w, bsz = is_y ? g.bin_h : g.bin_w;
    const bool first = (threadIdx.x & 1) == 0;
    const Lin1 l = lin_setup(first ? sample_coord(start, 0, bsz, 0, grid) : sample_coord(start, P - 1, bsz, grid - 1, grid), dim);
    box[threadIdx.x] = first ? l.lo : l.hi;
  }
  __syncthreads();
  const int y0 = box[0], y1 = box[1], x0 = box[2], x1 = box[3];
  float acc[P][P][2];
#pragma unroll
  for (int a = 0; a < P; ++a)
#pragma unroll
    for (int b = 0; b < P; ++b) acc[a][b][0] = acc[a][b][1] = 0.f;
  for (int py = y0; py <= y1; ++py) {
    const uint16_t* rowp = img + (long long)py * W * C;
    float row[P][2];
#pragma unroll
    for (int b = 0; b < P; ++b) row[b][0] = row[b][1] = 0.f;
    for (int xb = x0; xb <= x1; xb += 4) {
      uint32_t v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k)   // no branch inside a group of 4, so its loads go out back to back; columns past the
        v[k] = *reinterpret_cast<const uint32_t*>(rowp + (long long)min(xb + k, W - 1) * C);   // map re-read its last one, Wx = 0
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float4 wa = *reinterpret_cast<const float4*>(&wxT[xb + k][0]);
        const float4 wb = *reinterpret_cast<const float4*>(&wxT[xb + k][4]);
        const float w[P] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z};
        const float f0 = __uint_as_float(v[k] << 16), f1 = __uint_as_float(v[k] & 0xffff0000u);
#pragma unroll
        for (int b = 0; b < P; ++b) {
          row[b][0] += w[b] * f0;
          row[b][1] += w[b] * f1;
        }
      }
    }
    const float4 ya = *reinterpret_cast<const float4*>(&wyT[py][0]);
    const float4 yb = *reinterpret_cast<const float4*>(&wyT[py][4]);
    const float wy[P] = {ya.x, ya.y, ya.z, ya.w, yb.x, yb.y, yb.z};
#pragma unroll
    for (int a = 0; a < P; ++a) {
      if (wy[a] == 0.f) continue;   // uniform: a map row feeds the one or two row bins around it
#pragma unroll
      for (int b = 0; b < P; ++b) {
        acc[a][b][0] += wy[a] * row[b][0];
        acc[a][b][1] += wy[a] * row[b][1];
      }
    }
  }
  uint16_t* dst = out + (long long)n * P * P * C + c0;
#pragma unroll
  for (int a = 0; a < P; ++a)
#pragma unroll
    for (int b = 0; b < P; ++b)
      *reinterpret_cast<uint32_t*>(dst + (long long)(a * P + b) * C) = pack_bf16x2(acc[a][b][0], acc[a][b][1]);
}


// ---------------------------------------------------------------------------
// Forward, channels-last bf16, MATRIX form (the default for C % 128 == 0 and <= 64 bins).
// A bin is a Cartesian product of y samples and x samples, and a sample is dropped iff its y OR its x is out of
// range (ROIAlign_cuda.cu:22-31), so a roi's pooled map is a small matrix product over the pixels its samples touch:
//     out[bin][ch] = sum_p Wt[bin][p] * map[p][ch],     Wt[bin][(y, x)] = wy[bin_y][y] * wx[bin_x][x].
// One workgroup = one roi x 128 channels:
//   * the roi's pixel box (<= 96 pixels x 256 B) goes to LDS with 16-byte loads, once;
//   * one lane per bin derives the (at most 3 x 3) non-zero weights of its row in registers and scatters them into a
//     zeroed [64 bins][96 pixels] image, split into bf16 hi + lo parts (w = hi + lo to 2^-17: the MFMA takes bf16
//     operands and the weights are not bf16 numbers);
//   * v_mfma_f32_16x16x32_bf16 with M = channels (the A operand is the pixel image read TRANSPOSED with
//     ds_read_b64_tr_b16, rows XOR-swizzled so those reads are conflict-free), N = bins, K = pixels, f32 accumulators;
//   * the accumulators go through LDS once more so that every store is 16 bytes per lane and a bin row of the
//     workgroup is 256 contiguous bytes.
// The VALU forms this replaces were instruction-bound, not memory-bound: 9 pixels x 8 channels of unpack + FMA per
// lane and bin is 0.7 G lane-instructions per call (17 us of the chip's 64 lanes/clk/CU) for 38 MB of traffic.
// Covers sampling grids <= 2 x 2 with bins <= 2 pixels whose box fits (sampling_ratio 1-2, and adaptive grids of rois
// up to 14 pixels: every roi of a 14 x 14 map); other rois take the generic sample walk with loads straight from L2.
// f32 accumulation in a different order than the f32 NCHW kernel (feature-path tolerance; the bit-exact contract is
// the f32 kernel's).
// ---------------------------------------------------------------------------
constexpr int RM_MAXPIX = 96;     // staged pixels (K of the product): 96 x 256 B = 24 KiB
constexpr int RM_MAXBINS = 64;    // N of the product
constexpr int RM_WSTRIDE = 208;   // bytes per weight-image row: 96 bf16 + 16 (13 chunks: 16 rows hit 16 different chunks mod 16)
constexpr int RM_OSTRIDE = 272;   // bytes per output-image row: 128 bf16 + 16
constexpr int RM_THREADS = 256;   // 4 waves: 2 (channels) x 2 (bins)

typedef __attribute__((ext_vector_type(4))) short rm_s4;
typedef __attribute__((ext_vector_type(8))) short rm_s8;
typedef __attribute__((ext_vector_type(4))) float rm_f4;

// byte offset of 16-byte chunk `ch` (0..15) of pixel row `row` (cdna_hip_programming.md T10, image (b))
__device__ __forceinline__ int rm_poff(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

// The (at most three) consecutive pixels the one or two samples of a bin touch along one axis, with their summed
// weights (x 1 / grid): first = lo of sample 0; valid for grid <= 2 and sample spacing <= 1 pixel.
struct Axis3 {
  int first;
  float w[3];
};
// sample_coord for grid 1 or 2: x / grid == x * inv exactly (inv = 1 or 0.5), without the IEEE division sequence
__device__ __forceinline__ float sample_coord12(float start, int p, float bin, int i, float inv) {
  return start + (float)p * bin + (((float)i + .5f) * bin) * inv;
}
__device__ __forceinline__ Axis3 axis3(float start, int b, float bsz, int grid, int dim) {
  const float inv = grid == 2 ? 0.5f : 1.0f;
  const Lin1 s0 = lin_setup(sample_coord12(start, b, bsz, 0, inv), dim);
  const Lin1 s1 = lin_setup(sample_coord12(start, b, bsz, grid - 1, inv), dim);   // grid 1: the same sample again
  const float l0 = s0.valid ? s0.wl * inv : 0.f, h0 = s0.valid ? s0.wh * inv : 0.f;
  const bool two = grid > 1;
  const float l1 = (two && s1.valid) ? s1.wl * inv : 0.f, h1 = (two && s1.valid) ? s1.wh * inv : 0.f;
  const int d = s1.lo - s0.lo;   // 0 or 1 (2 only by float rounding of a spacing of exactly one pixel: then wh of sample 1 is ~0)
  Axis3 r;
  r.first = s0.lo;
  // sample 0: wl on first, wh on first + 1 (clamped at the border: hi == lo and wh == 0)
  // sample 1: wl on first + d, wh on first + d + 1
  r.w[0] = l0 + (d == 0 ? l1 : 0.f);
  r.w[1] = h0 + (d == 0 ? h1 : 0.f) + (d == 1 ? l1 : 0.f);
  r.w[2] = (d == 1 ? h1 : 0.f) + (d >= 2 ? l1 + h1 : 0.f);
  return r;
}

__global__ __launch_bounds__(RM_THREADS, 3) void roi_align_fwd_mfma_nhwc_bf16_kernel(
    const uint16_t* __restrict__ in, int B, int H, int W, int C, const float* __restrict__ rois, float scale, int PH,
    int PW, int sr, int slabs, uint16_t* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) unsigned char pimg[RM_MAXPIX * 256];            // pixels; later the output tile
  __shared__ __attribute__((aligned(16))) unsigned char wimg[2 * RM_MAXBINS * RM_WSTRIDE];  // weights: hi image, lo image
  // Workgroups go to the 8 XCDs round-robin by linear id and each XCD has its own L2: give XCD x the x-th contiguous
  // eighth of the (roi, slab group) units, so that rois sorted by image (as every caller here passes them) read
  // a map slice that stays in that XCD's L2
  int n, sg;
  {
    const int total = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
    const int xcd = lin & 7, per = total >> 3, rem = total & 7;
    const int unit = xcd * per + min(xcd, rem) + (lin >> 3);
    n = unit / (int)gridDim.x;
    sg = unit - n * (int)gridDim.x;
  }
  const int tid = threadIdx.x;
  const int c0 = sg * slabs * 128;   // this workgroup's `slabs` consecutive 128-channel slabs share the weight image
  const int nbins = PH * PW;
  const RoiGeom g = roi_geom(rois + (long long)n * 5, scale, PH, PW, sr, B);
  const uint16_t* img = in + (long long)g.batch * H * W * C + c0;
  uint16_t* obase = out + (long long)n * nbins * C + c0;
  // pixel box of the first / last sample of each axis (sample coordinates are monotonic; the clamped coordinates of
  // dropped samples only widen it); everything here is uniform over the workgroup
  const bool small = g.grid_h <= 2 && g.grid_w <= 2 && g.bin_h <= 2.0f && g.bin_w <= 2.0f;
  const float ivh = g.grid_h == 2 ? 0.5f : 1.0f, ivw = g.grid_w == 2 ? 0.5f : 1.0f;   // only used when `small`
  const Lin1 fy = lin_setup(sample_coord12(g.start_h, 0, g.bin_h, 0, ivh), H);
  const Lin1 ly = lin_setup(sample_coord12(g.start_h, PH - 1, g.bin_h, g.grid_h - 1, ivh), H);
  const Lin1 fx = lin_setup(sample_coord12(g.start_w, 0, g.bin_w, 0, ivw), W);
  const Lin1 lx = lin_setup(sample_coord12(g.start_w, PW - 1, g.bin_w, g.grid_w - 1, ivw), W);
  const int y0 = fy.lo, y1 = ly.hi, x0 = fx.lo, x1 = lx.hi;
  const int bw = x1 - x0 + 1, bh = y1 - y0 + 1, npix = bh * bw;
  const bool fast = small && npix <= RM_MAXPIX;
  if (!fast) {
    // larger sample grids / bins / boxes: the sample walk of the per-bin kernel, 16 lanes x 16 bytes per bin
    const int grp = tid >> 4, l = tid & 15;
    const float inv_count = 1.0f / g.count;
    for (int bs = grp; bs < nbins * slabs; bs += RM_THREADS / 16) {
      const int sl = bs / nbins, bin = bs - sl * nbins;
      const int a = bin / PW, b = bin - a * PW;
      const uint16_t* simg = img + sl * 128;
      float acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = 0.f;
      for (int iy = 0; iy < g.grid_h; ++iy) {
        const float y = sample_coord(g.start_h, a, g.bin_h, iy, g.grid_h);
        for (int ix = 0; ix < g.grid_w; ++ix) {
          const float x = sample_coord(g.start_w, b, g.bin_w, ix, g.grid_w);
          const Bilin bl = bilin_setup(y, x, H, W);
          if (!bl.valid) continue;
          const int pp[4] = {bl.p1, bl.p2, bl.p3, bl.p4};
          const float ww[4] = {bl.w1, bl.w2, bl.w3, bl.w4};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const uint4 v = *reinterpret_cast<const uint4*>(simg + (long long)pp[k] * C + l * 8);
            const uint32_t* q = &v.x;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              acc[2 * j] += ww[k] * __uint_as_float(q[j] << 16);
              acc[2 * j + 1] += ww[k] * __uint_as_float(q[j] & 0xffff0000u);
            }
          }
        }
      }
      uint4 o;
      o.x = pack_bf16x2(acc[0] * inv_count, acc[1] * inv_count);
      o.y = pack_bf16x2(acc[2] * inv_count, acc[3] * inv_count);
      o.z = pack_bf16x2(acc[4] * inv_count, acc[5] * inv_count);
      o.w = pack_bf16x2(acc[6] * inv_count, acc[7] * inv_count);
      *reinterpret_cast<uint4*>(obase + (long long)bin * C + sl * 128 + l * 8) = o;
    }
    return;
  }
  const int ksteps = (npix + 31) >> 5, kpad = ksteps * 32;
  // 1. the first slab's pixel loads in flight: 16 pixels x 16 chunks of 16 bytes per pass.  Rows npix .. kpad-1 are
  //    zero (0 x stale NaN = NaN).  One division per lane, then pixel p + 16 by increments.
  const int t16 = tid >> 4, tc = tid & 15;
  uint4 px[RM_MAXPIX / 16];
  int poff[RM_MAXPIX / 16];   // element offset of this lane's chunk of pixel p inside a slab; -1: no such pixel
  {
    // small-integer quotients through one v_rcp_f32: (a + .5) / bw is at least .5 / bw away from an integer
    const float rbw = __builtin_amdgcn_rcpf((float)bw);
    const int q16 = (int)(16.5f * rbw), r16 = 16 - q16 * bw;
    int py = (int)(((float)t16 + .5f) * rbw), pxx = t16 - py * bw;
#pragma unroll
    for (int it = 0; it < RM_MAXPIX / 16; ++it) {
      poff[it] = (t16 + 16 * it) < npix ? ((y0 + py) * W + x0 + pxx) * C + tc * 8 : -1;
      pxx += r16;
      py += q16;
      if (pxx >= bw) {
        pxx -= bw;
        ++py;
      }
      px[it] = make_uint4(0, 0, 0, 0);
      if (poff[it] >= 0) px[it] = *reinterpret_cast<const uint4*>(img + (n & (slabs - 1)) * 128 + poff[it]);
    }
  }
  // 2. zero the weight images
  for (int i = tid; i < 2 * RM_MAXBINS * RM_WSTRIDE / 16; i += RM_THREADS) reinterpret_cast<uint4*>(wimg)[i] = make_uint4(0, 0, 0, 0);
  __syncthreads();
  // 3. one lane per bin scatters its non-zero weights; zero-weight entries (the clamped duplicates at a border) go
  //    to the row's pad column, which no read touches
  if (tid < nbins) {
    const int a = (int)(((float)tid + .5f) * __builtin_amdgcn_rcpf((float)PW)), b = tid - a * PW;
    const Axis3 ay = axis3(g.start_h, a, g.bin_h, g.grid_h, H);
    const Axis3 ax = axis3(g.start_w, b, g.bin_w, g.grid_w, W);
    uint16_t* whi = reinterpret_cast<uint16_t*>(wimg + tid * RM_WSTRIDE);
    uint16_t* wlo = reinterpret_cast<uint16_t*>(wimg + RM_MAXBINS * RM_WSTRIDE + tid * RM_WSTRIDE);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const float w = ay.w[i] * ax.w[j];
        const int yy = ay.first + i, xx = ax.first + j;
        const int p = (w != 0.f && yy <= y1 && xx <= x1) ? (yy - y0) * bw + (xx - x0) : RM_MAXPIX;
        const uint16_t hi = f32_to_bf16(w);
        whi[p] = hi;
        wlo[p] = f32_to_bf16(w - bf16_to_f32(hi));
      }
  }
  // pixels -> image `dst`: rows XOR-swizzled by 16-byte chunk (rm_poff) and, on rows with bit 3 set, the two 8-byte
  // halves of a chunk swapped, so that the two 16-lane groups of a transposed read below land on different banks
  auto put_pixels = [&](unsigned char* dst) {
#pragma unroll
    for (int it = 0; it < RM_MAXPIX / 16; ++it) {
      const int p = t16 + 16 * it;
      if (p < kpad) {
        const bool sw = (p >> 3) & 1;
        uint4 v = px[it];
        if (sw) v = make_uint4(v.z, v.w, v.x, v.y);
        *reinterpret_cast<uint4*>(dst + rm_poff(p, tc)) = v;
      }
    }
  };
  // Slab order rotated by the roi index: the rois of one image run at the same time on one XCD, and only whoever
  // touches a slab of the map first waits for HBM; after the first step every workgroup reads what a neighbour fetched
  const int rot = n & (slabs - 1);   // slabs is 1, 2 or 4
  auto slab_of = [&](int i) { return (i + rot) & (slabs - 1); };
  auto get_pixels = [&](int i) {
    const int so = slab_of(i) * 128;
#pragma unroll
    for (int it = 0; it < RM_MAXPIX / 16; ++it)
      if (poff[it] >= 0) px[it] = *reinterpret_cast<const uint4*>(img + so + poff[it]);
  };
  put_pixels(pimg);
  if (slabs > 1) get_pixels(1);
  __syncthreads();
  // 4. out^T[ch][bin] = map^T[ch][p] . Wt^T[p][bin] on v_mfma_f32_16x16x32_bf16.  Waves 2 x 2: wave (wm, wn) owns
  //    channels 64 wm .. 64 wm + 63 and bins 32 wn .. 32 wn + 31.  Its weight fragments are the same for every slab and
  //    stay in registers, after which the weight image's LDS is the second pixel buffer.
  //    The A operand's 16 rows are the channels 32 pair + 8 (m >> 2) + 4 t + (m & 3) (tile t of a pair of M tiles): a lane
  //    then holds, over t = 0, 1, eight consecutive channels of one bin, a 16-byte store.
  const int lane = tid & 63, wv = tid >> 6, wm = wv & 1, wn = wv >> 1;
  const int kg = lane >> 4, li = lane & 15, q = li >> 2, pq = li & 3;
  rm_s8 wf[2][RM_MAXPIX / 32][2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int ks = 0; ks < RM_MAXPIX / 32; ++ks) {
      const unsigned char* wrow = wimg + (32 * wn + 16 * nt + li) * RM_WSTRIDE + (32 * ks + 8 * kg) * 2;
      wf[nt][ks][0] = *reinterpret_cast<const rm_s8*>(wrow);
      wf[nt][ks][1] = *reinterpret_cast<const rm_s8*>(wrow + RM_MAXBINS * RM_WSTRIDE);
    }
  __syncthreads();   // every wave holds its fragments: the weight image may be overwritten
  for (int sl = 0; sl < slabs; ++sl) {
    unsigned char* pcur = (sl & 1) ? wimg : pimg;
    if (sl + 1 < slabs) {
      put_pixels((sl & 1) ? pimg : wimg);   // loaded one slab ago
      if (sl + 2 < slabs) get_pixels(sl + 2);
    }
    rm_f4 acc[2][2][2];
#pragma unroll
    for (int pr = 0; pr < 2; ++pr)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[pr][t][nt] = rm_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < RM_MAXPIX / 32; ++ks) {
      if (ks < ksteps) {
        rm_s8 af[2][2];
#pragma unroll
        for (int pr = 0; pr < 2; ++pr)
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const int row = 32 * ks + 8 * kg + q, chunk = 8 * wm + 4 * pr + pq, half = 8 * (t ^ (kg & 1));
            const rm_s4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) rm_s4*)(pcur + rm_poff(row, chunk) + half));
            const rm_s4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) rm_s4*)(pcur + rm_poff(row + 4, chunk) + half));
            af[pr][t] = rm_s8{lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
          }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int pr = 0; pr < 2; ++pr)
#pragma unroll
              for (int t = 0; t < 2; ++t)
                acc[pr][t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[pr][t], wf[nt][ks][h], acc[pr][t][nt], 0, 0, 0);
      }
    }
    // accumulator rows 4 kg + r of tile t = channels 8 kg + 4 t + r of the pair; column li = bin
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int bin = 32 * wn + 16 * nt + li;
      if (bin < nbins) {
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          uint4 o;
          o.x = pack_bf16x2(acc[pr][0][nt][0], acc[pr][0][nt][1]);
          o.y = pack_bf16x2(acc[pr][0][nt][2], acc[pr][0][nt][3]);
          o.z = pack_bf16x2(acc[pr][1][nt][0], acc[pr][1][nt][1]);
          o.w = pack_bf16x2(acc[pr][1][nt][2], acc[pr][1][nt][3]);
          *reinterpret_cast<uint4*>(obase + (long long)bin * C + slab_of(sl) * 128 + 64 * wm + 32 * pr + 8 * kg) = o;
        }
      }
    }
    __syncthreads();   // this slab's image is read out, the next one is written
  }
}

static int grid_for(long long n) {
  long long b = (n + 255) / 256;
  if (b > 256LL * 32) b = 256LL * 32;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace trace

using namespace trace;

extern "C" int trace_roi_align_fwd_f32(const float* input, int B, int C, int H, int W,
                                       const float* rois, int K, float spatial_scale, int pooled_h,
                                       int pooled_w, int sampling_ratio, float* out,
                                       trace_stream_t stream) {
  TRACE_REQUIRE(B >= 0 && C >= 0 && H > 0 && W > 0 && K >= 0 && pooled_h > 0 && pooled_w > 0,
                "roi_align_fwd: bad shape B=%d C=%d H=%d W=%d K=%d", B, C, H, W, K);
  long long n = (long long)K * C * pooled_h * pooled_w;
  if (n == 0) return 0;  // ROIAlign_cuda.cu:278-281
  TRACE_REQUIRE(input && rois && out, "roi_align_fwd: null pointer");
  hipLaunchKernelGGL(roi_align_fwd_f32_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream,
                     n, input, spatial_scale, B, C, H, W, pooled_h, pooled_w, sampling_ratio, rois, out);
  return check_launch("roi_align_fwd_f32");
}

extern "C" int trace_roi_align_bwd_f32(const float* grad, const float* rois, int K,
                                       float spatial_scale, int pooled_h, int pooled_w, int B,
                                       int C, int H, int W, int sampling_ratio, float* grad_input,
                                       trace_stream_t stream) {
  TRACE_REQUIRE(B >= 0 && C >= 0 && H > 0 && W > 0 && K >= 0 && pooled_h > 0 && pooled_w > 0,
                "roi_align_bwd: bad shape");
  size_t gin_bytes = (size_t)B * C * H * W * sizeof(float);
  if (gin_bytes) {
    TRACE_REQUIRE(grad_input, "roi_align_bwd: null grad_input");
    hipError_t e = hipMemsetAsync(grad_input, 0, gin_bytes, (hipStream_t)stream);
    TRACE_REQUIRE(e == hipSuccess, "roi_align_bwd: memset: %s", hipGetErrorString(e));
  }
  long long n = (long long)K * C * pooled_h * pooled_w;
  if (n == 0) return 0;  // ROIAlign_cuda.cu:324-327
  TRACE_REQUIRE(grad && rois, "roi_align_bwd: null pointer");
  hipLaunchKernelGGL(roi_align_bwd_f32_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream,
                     n, grad, spatial_scale, B, C, H, W, pooled_h, pooled_w, sampling_ratio,
                     grad_input, rois);
  return check_launch("roi_align_bwd_f32");
}

extern "C" int trace_roi_pool_fwd_f32(const float* input, int B, int C, int H, int W,
                                      const float* rois, int K, float spatial_scale, int pooled_h,
                                      int pooled_w, float* out, int32_t* argmax,
                                      trace_stream_t stream) {
  TRACE_REQUIRE(B >= 0 && C >= 0 && H > 0 && W > 0 && K >= 0 && pooled_h > 0 && pooled_w > 0,
                "roi_pool_fwd: bad shape");
  long long n = (long long)K * C * pooled_h * pooled_w;
  if (n == 0) return 0;
  TRACE_REQUIRE(input && rois && out && argmax, "roi_pool_fwd: null pointer");
  hipLaunchKernelGGL(roi_pool_fwd_f32_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream,
                     n, input, spatial_scale, B, C, H, W, pooled_h, pooled_w, rois, out, argmax);
  return check_launch("roi_pool_fwd_f32");
}

extern "C" int trace_roi_pool_bwd_f32(const float* grad, const int32_t* argmax, const float* rois,
                                      int K, int pooled_h, int pooled_w, int B, int C, int H, int W,
                                      float* grad_input, trace_stream_t stream) {
  TRACE_REQUIRE(B >= 0 && C >= 0 && H > 0 && W > 0 && K >= 0, "roi_pool_bwd: bad shape");
  size_t gin_bytes = (size_t)B * C * H * W * sizeof(float);
  if (gin_bytes) {
    TRACE_REQUIRE(grad_input, "roi_pool_bwd: null grad_input");
    hipError_t e = hipMemsetAsync(grad_input, 0, gin_bytes, (hipStream_t)stream);
    TRACE_REQUIRE(e == hipSuccess, "roi_pool_bwd: memset: %s", hipGetErrorString(e));
  }
  long long n = (long long)K * C * pooled_h * pooled_w;
  if (n == 0) return 0;
  TRACE_REQUIRE(grad && argmax && rois, "roi_pool_bwd: null pointer");
  hipLaunchKernelGGL(roi_pool_bwd_f32_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream,
                     n, grad, argmax, B, C, H, W, pooled_h, pooled_w, grad_input, rois);
  return check_launch("roi_pool_bwd_f32");
}

// float64 forms of the four NCHW ops (see the kernels): same arguments, double data and rois
extern "C" int trace_roi_align_fwd_f64(const double* input, int B, int C, int H, int W, const double* rois, int K,
                                       float spatial_scale, int pooled_h, int pooled_w, int sampling_ratio,
                                       double* out, trace_stream_t stream) {
  TRACE_REQUIRE(B >= 0 && C >= 0 && H > 0 && W > 0 && K >= 0 && pooled_h > 0 && pooled_w > 0,
                "roi_align_fwd_f64: bad shape B=%d C=%d H=%d W=%d K=%d", B, C, H, W, K);
  long long n = (long long)K * C * pooled_h * pooled_w;
  if (n == 0) return 0;
  TRACE_REQUIRE(input && rois && out, "roi_align_fwd_f64: null pointer");
  hipLaunchKernelGGL(roi_align_fwd_f64_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream,
                     n, input, (double)spatial_scale, B, C, H, W, pooled_h, pooled_w, sampling_ratio, rois, out);
  return check_launch("roi_align_fwd_f64");
}

extern "C" int trace_roi_align_bwd_f64(const double* grad, const double* rois, int K, float spatial_scale,
                                       int pooled_h, int pooled_w, int B, int C, int H, int W, int sampling_ratio,
                                       double* grad_input, trace_stream_t stream) {
  TRACE_REQUIRE(B >= 0 && C >= 0 && H > 0 && W > 0 && K >= 0 && pooled_h > 0 && pooled_w > 0, "roi_align_bwd_f64: bad shape");
  size_t gin_bytes = (size_t)B * C * H * W * sizeof(double);
  if (gin_bytes) {
    TRACE_REQUIRE(grad_input, "roi_align_bwd_f64: null grad_input");
    hipError_t e = hipMemsetAsync(grad_input, 0, gin_bytes, (hipStream_t)stream);
    TRACE_REQUIRE(e == hipSuccess, "roi_align_bwd_f64: memset: %s", hipGetErrorString(e));
  }
  long long n = (long long)K * C * pooled_h * pooled_w;
  if (n == 0) return 0;
  TRACE_REQUIRE(grad && rois, "roi_align_bwd_f64: null pointer");
  hipLaunchKernelGGL(roi_align_bwd_f64_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream,
                     n, grad, (double)spatial_scale, B, C, H, W, pooled_h, pooled_w, sampling_ratio, grad_input, rois);
  return check_launch("roi_align_bwd_f64");
}

extern "C" int trace_roi_pool_fwd_f64(const double* input, int B, int C, int H, int W, const double* rois, int K,
                                      float spatial_scale, int pooled_h, int pooled_w, double* out, int32_t* argmax,
                                      trace_stream_t stream) {
  TRACE_REQUIRE(B >= 0 && C >= 0 && H > 0 && W > 0 && K >= 0 && pooled_h > 0 && pooled_w > 0, "roi_pool_fwd_f64: bad shape");
  long long n = (long long)K * C * pooled_h * pooled_w;
  if (n == 0) return 0;
  TRACE_REQUIRE(input && rois && out && argmax, "roi_pool_fwd_f64: null pointer");
  hipLaunchKernelGGL(roi_pool_fwd_f64_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream,
                     n, input, (double)spatial_scale, B, C, H, W, pooled_h, pooled_w, rois, out, argmax);
  return check_launch("roi_pool_fwd_f64");
}

extern "C" int trace_roi_pool_bwd_f64(const double* grad, const int32_t* argmax, const double* rois, int K,
                                      int pooled_h, int pooled_w, int B, int C, int H, int W, double* grad_input,
                                      trace_stream_t stream) {
  TRACE_REQUIRE(B >= 0 && C >= 0 && H > 0 && W > 0 && K >= 0, "roi_pool_bwd_f64: bad shape");
  size_t gin_bytes = (size_t)B * C * H * W * sizeof(double);
  if (gin_bytes) {
    TRACE_REQUIRE(grad_input, "roi_pool_bwd_f64: null grad_input");
    hipError_t e = hipMemsetAsync(grad_input, 0, gin_bytes, (hipStream_t)stream);
    TRACE_REQUIRE(e == hipSuccess, "roi_pool_bwd_f64: memset: %s", hipGetErrorString(e));
  }
  long long n = (long long)K * C * pooled_h * pooled_w;
  if (n == 0) return 0;
  TRACE_REQUIRE(grad && argmax && rois, "roi_pool_bwd_f64: null pointer");
  hipLaunchKernelGGL(roi_pool_bwd_f64_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream,
                     n, grad, argmax, B, C, H, W, pooled_h, pooled_w, grad_input, rois);
  return check_launch("roi_pool_bwd_f64");
}

extern "C" int trace_roi_align_fwd_nhwc_bf16(const uint16_t* input, int B, int H, int W, int C,
                                             const float* rois, int K, float spatial_scale,
                                             int pooled_h, int pooled_w, int sampling_ratio,
                                             uint16_t* out, trace_stream_t stream) {
  TRACE_REQUIRE(B >= 0 && C >= 0 && H > 0 && W > 0 && K >= 0 && pooled_h > 0 && pooled_w > 0,
                "roi_align_fwd_nhwc: bad shape");
  TRACE_REQUIRE(C % 8 == 0, "roi_align_fwd_nhwc: C=%d must be a multiple of 8", C);
  if (K == 0 || C == 0) return 0;
  TRACE_REQUIRE(K <= 65535, "roi_align_fwd_nhwc: K=%d exceeds 65535 rois per call", K);
  TRACE_REQUIRE(input && rois && out, "roi_align_fwd_nhwc: null pointer");
  static const int roi_alg = getenv("TRACE_ROI_ALG") ? atoi(getenv("TRACE_ROI_ALG")) : 0;   // A/B: 1 = sep7, 2 = per-bin
  const bool no_sep = roi_alg == 2;
  if (roi_alg == 0 && C % 128 == 0 && pooled_h * pooled_w <= RM_MAXBINS) {
    // a workgroup builds the roi's weight image once and walks `slabs` 128-channel slabs with it, as long as the grid
    // still holds two workgroups per CU
    const int nslab = C / 128;
    int slabs = 1;
    for (int sgl = 4; sgl > 1; sgl >>= 1)
      if (nslab % sgl == 0 && (long long)(nslab / sgl) * K >= 512) { slabs = sgl; break; }
    hipLaunchKernelGGL(roi_align_fwd_mfma_nhwc_bf16_kernel, dim3(nslab / slabs, K), dim3(RM_THREADS), 0, (hipStream_t)stream, input, B, H,
                       W, C, rois, spatial_scale, pooled_h, pooled_w, sampling_ratio, slabs, out);
    return check_launch("roi_align_fwd_mfma_nhwc_bf16");
  }
  // C % 128 != 0 or more than 64 bins.  Measured on the object-head call (160 rois x 2048 ch, 14 x 14 maps, graph of 20
  // launches): per-bin 37.4 us, separable 26.6 us, matrix form above 16.6 us
  if (!no_sep && sampling_ratio >= 2 && pooled_h == 7 && pooled_w == 7 && H <= RS_MAX_DIM && W <= RS_MAX_DIM && C % 128 == 0) {
    hipLaunchKernelGGL(roi_align_fwd_sep7_nhwc_bf16_kernel, dim3(C / 128, K), dim3(64), 0, (hipStream_t)stream,
                       input, B, H, W, C, rois, spatial_scale, sampling_ratio, out);
    return check_launch("roi_align_fwd_sep7_nhwc_bf16");
  }
  int threads = C / 8 >= 256 ? 256 : ((C / 8 + 63) / 64) * 64;
  hipLaunchKernelGGL(roi_align_fwd_nhwc_bf16_kernel, dim3(pooled_h * pooled_w, K), dim3(threads), 0,
                     (hipStream_t)stream, input, B, H, W, C, rois, spatial_scale, pooled_h, pooled_w,
                     sampling_ratio, out);
  return check_launch("roi_align_fwd_nhwc_bf16");
}

extern "C" int trace_roi_align_avg_nhwc_bf16(const uint16_t* input, int B, int H, int W, int C,
                                             const float* rois, int K, float spatial_scale,
                                             int pooled_h, int pooled_w, int sampling_ratio,
                                             uint16_t* out, trace_stream_t stream) {
  TRACE_REQUIRE(B >= 0 && C >= 0 && H > 0 && W > 0 && K >= 0 && pooled_h > 0 && pooled_w > 0,
                "roi_align_avg_nhwc: bad shape");
  TRACE_REQUIRE(C % 8 == 0, "roi_align_avg_nhwc: C=%d must be a multiple of 8", C);
  if (K == 0 || C == 0) return 0;
  TRACE_REQUIRE(input && rois && out, "roi_align_avg_nhwc: null pointer");
  TRACE_REQUIRE(H <= RA_MAX_DIM && W <= RA_MAX_DIM, "roi_align_avg_nhwc: map %dx%d exceeds %d", H, W, RA_MAX_DIM);
  int threads = C / 8 >= 256 ? 256 : ((C / 8 + 63) / 64) * 64;
  hipLaunchKernelGGL(roi_align_avg_nhwc_bf16_kernel, dim3(K), dim3(threads), 0, (hipStream_t)stream,
                     input, B, H, W, C, rois, spatial_scale, pooled_h, pooled_w, sampling_ratio, out);
  return check_launch("roi_align_avg_nhwc_bf16");
}
