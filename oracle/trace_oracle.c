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

/* ---- ROIAlign / ROIPool: roi_oracle.inc, once per floating type */
#define REAL float
#define FN(name) name
#define R_FMAX fmaxf
#define R_CEIL ceilf
#define R_ROUND roundf
#define R_FLOOR floorf
#include "roi_oracle.inc"
#undef REAL
#undef FN
#undef R_FMAX
#undef R_CEIL
#undef R_ROUND
#undef R_FLOOR
#undef CLAMP
#define REAL double
#define FN(name) name##_f64
#define R_FMAX fmax
#define R_CEIL ceil
#define R_ROUND round
#define R_FLOOR floor
#include "roi_oracle.inc"
#undef REAL
#undef FN
#undef R_FMAX
#undef R_CEIL
#undef R_ROUND
#undef R_FLOOR
#undef CLAMP

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
