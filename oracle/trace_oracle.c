/*
 * TEST INFRASTRUCTURE ONLY — CPU oracle for the TRACE hot path (plain C).
 *
 * A restatement of the reference's algorithms, used by tests/, by
 * __graft_entry__.smoke() and by bench.py's cpu_baseline leg as the checker /
 * CPU baseline.  The product (trace_amd/) never links, loads or calls this.
 *
 * Pinned against the reference itself: oracle/_ref (the reference's own
 * csrc/cpu sources compiled by oracle/build_ref.py) for roi_align forward and
 * nms; see tests/test_oracle_pinning.py and tests/golden/.
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC (see oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- ROIAlign forward: Detectron_pytorch/lib/model/csrc/cpu/ROIAlign_cpu.cpp:17-219 */
typedef struct {
  int pos1, pos2, pos3, pos4;
  float w1, w2, w3, w4;
  int valid;
} precalc_t;

/* ROIAlign_cpu.cpp:17-111 */
static void pre_calc(int height, int width, int pooled_h, int pooled_w, int iy_upper, int ix_upper,
                     float roi_start_h, float roi_start_w, float bin_size_h, float bin_size_w,
                     int grid_h, int grid_w, precalc_t* pc) {
  int idx = 0;
  for (int ph = 0; ph < pooled_h; ph++)
    for (int pw = 0; pw < pooled_w; pw++)
      for (int iy = 0; iy < iy_upper; iy++) {
        const float yy = roi_start_h + ph * bin_size_h + (float)(iy + .5f) * bin_size_h / (float)grid_h;
        for (int ix = 0; ix < ix_upper; ix++) {
          const float xx = roi_start_w + pw * bin_size_w + (float)(ix + .5f) * bin_size_w / (float)grid_w;
          float x = xx, y = yy;
          precalc_t p;
          if (y < -1.0 || y > height || x < -1.0 || x > width) { /* :47-61 */
            memset(&p, 0, sizeof(p));
            pc[idx++] = p;
            continue;
          }
          if (y <= 0) y = 0;
          if (x <= 0) x = 0;
          int y_low = (int)y, x_low = (int)x, y_high, x_high;
          if (y_low >= height - 1) { y_high = y_low = height - 1; y = (float)y_low; } else { y_high = y_low + 1; }
          if (x_low >= width - 1) { x_high = x_low = width - 1; x = (float)x_low; } else { x_high = x_low + 1; }
          float ly = y - y_low, lx = x - x_low;
          float hy = (float)(1. - ly), hx = (float)(1. - lx);
          p.w1 = hy * hx; p.w2 = hy * lx; p.w3 = ly * hx; p.w4 = ly * lx;
          p.pos1 = y_low * width + x_low; p.pos2 = y_low * width + x_high;
          p.pos3 = y_high * width + x_low; p.pos4 = y_high * width + x_high;
          p.valid = 1;
          pc[idx++] = p;
        }
      }
}

static void roi_geometry(const float* r, float scale, int ph, int pw, int sr, int* batch,
                         float* start_w, float* start_h, float* bin_w, float* bin_h, int* grid_h,
                         int* grid_w) {
  /* ROIAlign_cpu.cpp:138-166 */
  *batch = (int)r[0];
  *start_w = r[1] * scale;
  *start_h = r[2] * scale;
  float end_w = r[3] * scale, end_h = r[4] * scale;
  float roi_w = fmaxf(end_w - *start_w, 1.f), roi_h = fmaxf(end_h - *start_h, 1.f);
  *bin_h = roi_h / (float)ph;
  *bin_w = roi_w / (float)pw;
  *grid_h = sr > 0 ? sr : (int)ceilf(roi_h / ph);
  *grid_w = sr > 0 ? sr : (int)ceilf(roi_w / pw);
}

void oracle_roi_align_fwd(const float* in, int B, int C, int H, int W, const float* rois, int K,
                          float scale, int PH, int PW, int sr, float* out) {
  (void)B;
  for (int n = 0; n < K; n++) {
    int b, gh, gw;
    float sw, sh, bw, bh;
    roi_geometry(rois + n * 5, scale, PH, PW, sr, &b, &sw, &sh, &bw, &bh, &gh, &gw);
    const float count = (float)(gh * gw);
    precalc_t* pc = (precalc_t*)malloc(sizeof(precalc_t) * (size_t)(gh * gw * PH * PW + 1));
    pre_calc(H, W, PH, PW, gh, gw, sh, sw, bh, bw, gh, gw, pc);
    for (int c = 0; c < C; c++) { /* :190-217 */
      const float* plane = in + ((size_t)b * C + c) * H * W;
      int pi = 0;
      for (int ph = 0; ph < PH; ph++)
        for (int pw = 0; pw < PW; pw++) {
          float v = 0.f;
          for (int iy = 0; iy < gh; iy++)
            for (int ix = 0; ix < gw; ix++) {
              precalc_t p = pc[pi++];
              v += p.w1 * plane[p.pos1] + p.w2 * plane[p.pos2] + p.w3 * plane[p.pos3] + p.w4 * plane[p.pos4];
            }
          v /= count;
          out[(((size_t)n * C + c) * PH + ph) * PW + pw] = v;
        }
    }
    free(pc);
  }
}

/* ---- ROIAlign backward: csrc/cuda/ROIAlign_cuda.cu:177-254 (serial, in index order) */
void oracle_roi_align_bwd(const float* grad, const float* rois, int K, float scale, int PH, int PW,
                          int B, int C, int H, int W, int sr, float* gin) {
  memset(gin, 0, sizeof(float) * (size_t)B * C * H * W);
  for (int n = 0; n < K; n++) {
    int b, gh, gw;
    float sw, sh, bw, bh;
    roi_geometry(rois + n * 5, scale, PH, PW, sr, &b, &sw, &sh, &bw, &bh, &gh, &gw);
    const float count = (float)(gh * gw);
    precalc_t* pc = (precalc_t*)malloc(sizeof(precalc_t) * (size_t)(gh * gw * PH * PW + 1));
    pre_calc(H, W, PH, PW, gh, gw, sh, sw, bh, bw, gh, gw, pc);
    for (int c = 0; c < C; c++) {
      float* plane = gin + ((size_t)b * C + c) * H * W;
      int pi = 0;
      for (int ph = 0; ph < PH; ph++)
        for (int pw = 0; pw < PW; pw++) {
          float top = grad[(((size_t)n * C + c) * PH + ph) * PW + pw];
          for (int iy = 0; iy < gh; iy++)
            for (int ix = 0; ix < gw; ix++) {
              precalc_t p = pc[pi++];
              if (!p.valid) continue; /* :244 */
              plane[p.pos1] += top * p.w1 / count; /* :239-249 */
              plane[p.pos2] += top * p.w2 / count;
              plane[p.pos3] += top * p.w3 / count;
              plane[p.pos4] += top * p.w4 / count;
            }
        }
    }
    free(pc);
  }
}

/* ---- ROIPool forward/backward: csrc/cuda/ROIPool_cuda.cu:16-108 */
void oracle_roi_pool_fwd(const float* in, int B, int C, int H, int W, const float* rois, int K,
                         float scale, int PH, int PW, float* out, int32_t* argmax) {
  (void)B;
  for (int n = 0; n < K; n++) {
    const float* r = rois + n * 5;
    int b = (int)r[0];
    int rsw = (int)roundf(r[1] * scale), rsh = (int)roundf(r[2] * scale);
    int rew = (int)roundf(r[3] * scale), reh = (int)roundf(r[4] * scale);
    int rw = rew - rsw + 1 > 1 ? rew - rsw + 1 : 1;
    int rh = reh - rsh + 1 > 1 ? reh - rsh + 1 : 1;
    float bin_h = (float)rh / (float)PH, bin_w = (float)rw / (float)PW;
    for (int c = 0; c < C; c++) {
      const float* plane = in + ((size_t)b * C + c) * H * W;
      for (int ph = 0; ph < PH; ph++)
        for (int pw = 0; pw < PW; pw++) {
          int hstart = (int)floorf((float)ph * bin_h), wstart = (int)floorf((float)pw * bin_w);
          int hend = (int)ceilf((float)(ph + 1) * bin_h), wend = (int)ceilf((float)(pw + 1) * bin_w);
#define CLAMP(v, lo, hi) ((v) < (lo) ? (lo) : ((v) > (hi) ? (hi) : (v)))
          hstart = CLAMP(hstart + rsh, 0, H); hend = CLAMP(hend + rsh, 0, H);
          wstart = CLAMP(wstart + rsw, 0, W); wend = CLAMP(wend + rsw, 0, W);
          int empty = (hend <= hstart) || (wend <= wstart);
          float maxval = empty ? 0.f : -3.402823466e+38f;
          int maxidx = -1;
          for (int h = hstart; h < hend; h++)
            for (int w = wstart; w < wend; w++)
              if (plane[h * W + w] > maxval) { maxval = plane[h * W + w]; maxidx = h * W + w; }
          size_t o = (((size_t)n * C + c) * PH + ph) * PW + pw;
          out[o] = maxval;
          argmax[o] = maxidx;
        }
    }
  }
}

void oracle_roi_pool_bwd(const float* grad, const int32_t* argmax, const float* rois, int K, int PH,
                         int PW, int B, int C, int H, int W, float* gin) {
  memset(gin, 0, sizeof(float) * (size_t)B * C * H * W);
  for (int n = 0; n < K; n++) {
    int b = (int)rois[n * 5];
    for (int c = 0; c < C; c++)
      for (int i = 0; i < PH * PW; i++) {
        size_t o = ((size_t)n * C + c) * PH * PW + i;
        if (argmax[o] != -1) gin[((size_t)b * C + c) * H * W + argmax[o]] += grad[o];
      }
  }
}

/* ---- NMS: Detectron_pytorch/lib/utils/cython_nms.pyx:37-87, csrc/cpu/nms_cpu.cpp:5-65.
 * mode 0: suppress if ovr >= thresh (Cython / CPU); mode 1: ovr > thresh (cuda/nms.cu:60).
 * Order: descending score; ties broken by ascending index (the reference leaves
 * ties to the sort implementation; tests use distinct scores for pinning).
 * Returns the number kept; keep[] holds ascending original indices. */
typedef struct { float s; int i; } si_t;
static int cmp_si(const void* a, const void* b) {
  const si_t* x = (const si_t*)a; const si_t* y = (const si_t*)b;
  if (x->s > y->s) return -1;
  if (x->s < y->s) return 1;
  return x->i < y->i ? -1 : (x->i > y->i ? 1 : 0);
}
int oracle_nms(const float* boxes, const float* scores, int N, float thresh, int mode, int64_t* keep) {
  if (N <= 0) return 0;
  si_t* ord = (si_t*)malloc(sizeof(si_t) * N);
  float* areas = (float*)malloc(sizeof(float) * N);
  uint8_t* sup = (uint8_t*)calloc(N, 1);
  for (int i = 0; i < N; i++) {
    ord[i].s = scores[i]; ord[i].i = i;
    areas[i] = (boxes[i * 4 + 2] - boxes[i * 4 + 0] + 1) * (boxes[i * 4 + 3] - boxes[i * 4 + 1] + 1);
  }
  qsort(ord, N, sizeof(si_t), cmp_si);
  for (int _i = 0; _i < N; _i++) {
    int i = ord[_i].i;
    if (sup[i]) continue;
    float ix1 = boxes[i * 4], iy1 = boxes[i * 4 + 1], ix2 = boxes[i * 4 + 2], iy2 = boxes[i * 4 + 3];
    float iarea = areas[i];
    for (int _j = _i + 1; _j < N; _j++) {
      int j = ord[_j].i;
      if (sup[j]) continue;
      float xx1 = fmaxf(ix1, boxes[j * 4]), yy1 = fmaxf(iy1, boxes[j * 4 + 1]);
      float xx2 = fminf(ix2, boxes[j * 4 + 2]), yy2 = fminf(iy2, boxes[j * 4 + 3]);
      float w = fmaxf(0.0f, xx2 - xx1 + 1), h = fmaxf(0.0f, yy2 - yy1 + 1);
      float inter = w * h;
      float ovr = inter / (iarea + areas[j] - inter);
      if (mode == 1 ? (ovr > thresh) : (ovr >= thresh)) sup[j] = 1;
    }
  }
  int k = 0;
  for (int i = 0; i < N; i++) if (!sup[i]) keep[k++] = i;
  free(ord); free(areas); free(sup);
  return k;
}

/* ---- cython_bbox.bbox_overlaps: Detectron_pytorch/lib/utils/cython_bbox.pyx:32-73 */
void oracle_bbox_overlaps(const float* boxes, int N, const float* query, int K, float* out) {
  for (int k = 0; k < K; k++) {
    const float* q = query + k * 4;
    /* Cython turns the integer literal beside a float32 operand into the C double 1.0: differences are float32, "+ 1",
       the products and the sum of `ua` run in double and are rounded to float32 on assignment (pinned by
       tests/golden/cy_overlaps.npz, the compiled .pyx's own outputs; an all-float32 restatement is 1 ulp off in 1-5 %
       of the entries) */
    float box_area = (float)(((double)(q[2] - q[0]) + 1.0) * ((double)(q[3] - q[1]) + 1.0));
    for (int n = 0; n < N; n++) {
      const float* b = boxes + n * 4;
      float o = 0.f;
      float iw = (float)((double)(fminf(b[2], q[2]) - fmaxf(b[0], q[0])) + 1.0);
      if (iw > 0) {
        float ih = (float)((double)(fminf(b[3], q[3]) - fmaxf(b[1], q[1])) + 1.0);
        if (ih > 0) {
          float ua = (float)(((double)(b[2] - b[0]) + 1.0) * ((double)(b[3] - b[1]) + 1.0) + (double)box_area - (double)(iw * ih));
          o = iw * ih / ua;
        }
      }
      out[(size_t)n * K + k] = o;
    }
  }
}

/* ---- cython_bbox_rel.bbox_pair_overlaps: lib/utils_rel/cython_bbox_rel.pyx:37-77 */
void oracle_bbox_pair_overlaps(const float* b1, const float* b2, int N, float* out) {
  for (int n = 0; n < N; n++) {
    const float* a = b1 + n * 4; const float* q = b2 + n * 4;
    float box_area = (float)(((double)(q[2] - q[0]) + 1.0) * ((double)(q[3] - q[1]) + 1.0));   /* as above */
    float o = 0.f;
    float iw = (float)((double)(fminf(a[2], q[2]) - fmaxf(a[0], q[0])) + 1.0);
    if (iw > 0) {
      float ih = (float)((double)(fminf(a[3], q[3]) - fmaxf(a[1], q[1])) + 1.0);
      if (ih > 0) {
        float ua = (float)(((double)(a[2] - a[0]) + 1.0) * ((double)(a[3] - a[1]) + 1.0) + (double)box_area - (double)(iw * ih));
        o = iw * ih / ua;
      }
    }
    out[n] = o;
  }
}
