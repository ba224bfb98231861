/*
 * trace_amd.h — C-ABI of libtrace_amd.so: the MI355X (gfx950) implementation of
 * TRACE's per-clip video-relation hot path.
 *
 * Conventions:
 *   - every entry point returns 0 on success, nonzero on error; the message is
 *     available from trace_last_error() (thread-local);
 *   - all data pointers are DEVICE pointers owned by the caller; nothing is
 *     allocated inside (workspaces are passed in, sizes come from the
 *     trace_*_workspace_bytes helpers); no host synchronisation;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - bf16 tensors are passed as `const uint16_t*` (raw bfloat16 bits).
 *
 * Each declaration cites the reference interface (file:line under
 * /root/reference) it replaces.
 */
#ifndef TRACE_AMD_H
#define TRACE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* trace_stream_t;

/* Library/version probes. */
const char* trace_last_error(void);
int trace_abi_version(void);

/* ------------------------------------------------------------------------ */
/* ROIAlign / ROIPool: Detectron_pytorch/lib/model/csrc/vision.cpp:7-13      */
/* ------------------------------------------------------------------------ */

/* _C.roi_align_forward — csrc/ROIAlign.h:11-25, cuda/ROIAlign_cuda.cu:64-122,
 * cpu/ROIAlign_cpu.cpp:113-219.  input (B,C,H,W) f32 NCHW, rois (K,5)
 * [batch,x1,y1,x2,y2] f32, out (K,C,PH,PW) f32.  sampling_ratio<=0 = adaptive
 * ceil(roi/pooled) grid.  Arithmetic order follows the CPU reference (no FMA
 * contraction) so results are bit-identical to it. */
int trace_roi_align_fwd_f32(const float* input, int B, int C, int H, int W,
                            const float* rois, int K, float spatial_scale,
                            int pooled_h, int pooled_w, int sampling_ratio,
                            float* out, trace_stream_t stream);

/* _C.roi_align_backward — csrc/ROIAlign.h:27-45, cuda/ROIAlign_cuda.cu:177-254,
 * 302-346.  grad (K,C,PH,PW) f32 -> grad_input (B,C,H,W) f32 (zeroed here, on
 * `stream`, then accumulated with float atomics as the reference does). */
int trace_roi_align_bwd_f32(const float* grad, const float* rois, int K,
                            float spatial_scale, int pooled_h, int pooled_w,
                            int B, int C, int H, int W, int sampling_ratio,
                            float* grad_input, trace_stream_t stream);

/* _C.roi_pool_forward — csrc/ROIPool.h:11-25, cuda/ROIPool_cuda.cu:16-77.
 * out (K,C,PH,PW) f32, argmax (K,C,PH,PW) int32 (index into the H*W plane,
 * -1 when the bin is empty). */
int trace_roi_pool_fwd_f32(const float* input, int B, int C, int H, int W,
                           const float* rois, int K, float spatial_scale,
                           int pooled_h, int pooled_w, float* out,
                           int32_t* argmax, trace_stream_t stream);

/* _C.roi_pool_backward — csrc/ROIPool.h:27-46, cuda/ROIPool_cuda.cu:79-108. */
int trace_roi_pool_bwd_f32(const float* grad, const int32_t* argmax,
                           const float* rois, int K, int pooled_h,
                           int pooled_w, int B, int C, int H, int W,
                           float* grad_input, trace_stream_t stream);

/* The double half of the reference's dtype dispatch (AT_DISPATCH_FLOATING_TYPES: cuda/ROIAlign_cuda.cu:283,330,
 * cuda/ROIPool_cuda.cu:129,178, cpu/ROIAlign_cpu.cpp:243): the four ops above with float64 data and float64 rois
 * (the reference reads the rois in the input's dtype), same arguments, same rules. */
int trace_roi_align_fwd_f64(const double* input, int B, int C, int H, int W,
                            const double* rois, int K, float spatial_scale,
                            int pooled_h, int pooled_w, int sampling_ratio,
                            double* out, trace_stream_t stream);
int trace_roi_align_bwd_f64(const double* grad, const double* rois, int K,
                            float spatial_scale, int pooled_h, int pooled_w,
                            int B, int C, int H, int W, int sampling_ratio,
                            double* grad_input, trace_stream_t stream);
int trace_roi_pool_fwd_f64(const double* input, int B, int C, int H, int W,
                           const double* rois, int K, float spatial_scale,
                           int pooled_h, int pooled_w, double* out,
                           int32_t* argmax, trace_stream_t stream);
int trace_roi_pool_bwd_f64(const double* grad, const int32_t* argmax,
                           const double* rois, int K, int pooled_h,
                           int pooled_w, int B, int C, int H, int W,
                           double* grad_input, trace_stream_t stream);

/* MI355X-native layout variant of roi_align_forward used inside the fused
 * pipeline: input (B,H,W,C) bf16 channels-last, out (K,PH,PW,C) bf16; same
 * sampling rule as above, f32 accumulation.  Replaces the per-call
 * `torch.from_numpy(rois).cuda()` + NCHW gather at
 * lib/modeling_rel/model_builder_rel.py:1389-1394 and
 * lib/combine_tree/cluster_tree.py:729-732.  C must be a multiple of 8. */
int trace_roi_align_fwd_nhwc_bf16(const uint16_t* input, int B, int H, int W,
                                  int C, const float* rois, int K,
                                  float spatial_scale, int pooled_h,
                                  int pooled_w, int sampling_ratio,
                                  uint16_t* out, trace_stream_t stream);

/* ROIAlign fused with the AdaptiveAvgPool2d((1,1)) that follows it in
 * Tail_Layer_3D.map2vec (lib/i3d/box_head3d.py:59-61, I3D_DC5): out (K,C) bf16 =
 * mean over the pooled_h*pooled_w bins; the (K,PH,PW,C) tensor is never written. */
int trace_roi_align_avg_nhwc_bf16(const uint16_t* input, int B, int H, int W,
                                  int C, const float* rois, int K,
                                  float spatial_scale, int pooled_h,
                                  int pooled_w, int sampling_ratio,
                                  uint16_t* out, trace_stream_t stream);

/* ------------------------------------------------------------------------ */
/* NMS: csrc/nms.h:10-28 (_C.nms), cpu/nms_cpu.cpp:5-65, cuda/nms.cu:23-131, */
/*      and the run-time path Detectron_pytorch/lib/utils/cython_nms.pyx:37-87 */
/* ------------------------------------------------------------------------ */
#define TRACE_NMS_GE 0 /* suppress if IoU >= thresh (Cython / CPU reference) */
#define TRACE_NMS_GT 1 /* suppress if IoU >  thresh (CUDA reference)          */

size_t trace_nms_workspace_bytes(int n_boxes);

/* boxes (N,4) f32 [x1,y1,x2,y2], scores (N,) f32.  Greedy NMS in descending
 * score order (ties: lower index first), "+1" areas, f32 IoU exactly as the
 * reference.  keep_out receives the kept ORIGINAL indices in ascending order
 * (cython_nms.pyx:87, nms.cu:127-130), *num_keep (device int32) their count. */
int trace_nms(const float* boxes, const float* scores, int N, float thresh,
              int mode, int64_t* keep_out, int32_t* num_keep, void* workspace,
              size_t workspace_bytes, trace_stream_t stream);

/* Per-class NMS of one image in ONE pass — lib/modeling_rel/model_builder_rel.py:1286-1313
 * (the `for j in range(1, num_classes)` loop around box_utils.nms): boxes of different `group`
 * (class) never suppress each other; within a group the result equals trace_nms on that group's
 * boxes alone.  keep_out: kept ORIGINAL indices ascending.  Group ids must lie in [0, 2^18 - 2] (class ids): the
 * boxes are sorted by (group, score) so that the groups are reduced side by side, one short chain each. */
int trace_nms_grouped(const float* boxes, const float* scores, const int32_t* group,
                      int N, float thresh, int mode, int64_t* keep_out,
                      int32_t* num_keep, void* workspace, size_t workspace_bytes,
                      trace_stream_t stream);

/* trace_nms for callers that cannot allocate (code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc): NO counterpart in the reference.  Same result as trace_nms, but the scratch is a
 * 2.13 MiB array inside the library's device image (one copy per GPU, kept until the process ends) instead of
 * a caller workspace; calls no hipMalloc / hipFree or other allocating routine.  N <= 4096 boxes (error
 * otherwise).  All calls on one GPU share that array: calls that overlap on one GPU (two streams, two
 * threads) race - order them on one stream. */
int trace_nms_noalloc(const float* boxes, const float* scores, int N, float thresh, int mode,
                      int64_t* keep_out, int32_t* num_keep, trace_stream_t stream);

/* ------------------------------------------------------------------------ */
/* Box helpers                                                               */
/* ------------------------------------------------------------------------ */

/* utils.fpn.map_rois_to_fpn_levels — Detectron_pytorch/lib/utils/fpn.py:11-28
 * (+ boxes_area, utils/boxes.py:58-73).  rois (N,4) f32 -> levels (N,) f32. */
int trace_fpn_levels(const float* boxes, int N, int k_min, int k_max,
                     float canonical_scale, float canonical_level,
                     float* levels, trace_stream_t stream);

/* utils.boxes.bbox_transform + clip_tiled_boxes — Detectron_pytorch/lib/utils/boxes.py:160-200,
 * 142-157 (called from model_builder_rel.py:1261-1278 get_det_boxes).  boxes (R,4), deltas (R,4K)
 * -> out (R,4K) decoded with weights (wx,wy,ww,wh), dw/dh clipped at xform_clip
 * (cfg.BBOX_XFORM_CLIP = log(1000/16)), clipped to [0,im_w-1] x [0,im_h-1]. */
int trace_bbox_transform_clip(const float* boxes, const float* deltas, int R, int K,
                              float wx, float wy, float ww, float wh,
                              float xform_clip, float im_h, float im_w, float* out,
                              trace_stream_t stream);

/* cython_bbox.bbox_overlaps — Detectron_pytorch/lib/utils/cython_bbox.pyx:32-73.
 * boxes (N,4), query (K,4) -> overlaps (N,K) f32. */
int trace_bbox_overlaps(const float* boxes, int N, const float* query, int K,
                        float* overlaps, trace_stream_t stream);

/* cython_bbox_rel.bbox_pair_overlaps — lib/utils_rel/cython_bbox_rel.pyx:37-77. */
int trace_bbox_pair_overlaps(const float* boxes1, const float* boxes2, int N,
                             float* overlaps, trace_stream_t stream);

/* RelPN pair enumeration + union rois — lib/modeling_rel/relpn_heads.py:79-84,
 * 39-46, 117 and lib/utils_rel/boxes_rel.py:100-106.  det_rois (n,5) ->
 * sbj_inds/obj_inds (P,) int32, rel_rois (P,5) f32, P = n*(n-1) (n if n==1). */
int trace_relpn_pairs(const float* det_rois, int n, int32_t* sbj_inds,
                      int32_t* obj_inds, float* rel_rois,
                      trace_stream_t stream);

/* ------------------------------------------------------------------------ */
/* I3D backbone: lib/i3d/resnet3d.py:155-318, 718-802 (+ mmcv ConvModule /   */
/* NonLocal3d call sites resnet3d.py:95,250,262,275,289,549,721)             */
/* ------------------------------------------------------------------------ */

/* (N,C,T,H,W) or (N,T,C,H,W) f32 clip -> channels-last bf16
 * (N, pt0+T+pt1, ph0+H+ph1, pw0+W+pw1, Cp): channel dim zero-padded to Cp and a
 * zero halo written around T/H/W so that the stem conv needs no bounds checks;
 * replaces the permute at lib/modeling_rel/model_builder_rel.py:978.
 * in_ntchw!=0 selects the (N,T,C,H,W) input order (the layout all_frames
 * arrives in, :688). */
int trace_clip_to_ndhwc_bf16(const float* in, int N, int C, int T, int H,
                             int W, int in_ntchw, int Cp, int pt0, int pt1,
                             int ph0, int ph1, int pw0, int pw1, uint16_t* out,
                             trace_stream_t stream);

#define TRACE_EPI_RELU 1     /* ReLU after bias(+residual)                   */
#define TRACE_EPI_RESIDUAL 2 /* add `residual` (same shape as out) first     */
#define TRACE_EPI_OUT_F32 4  /* write f32 instead of bf16                    */
#define TRACE_EPI_LEAKY 8    /* LeakyReLU(0.1) instead of ReLU               */
#define TRACE_EPI_RES_POST 16 /* add `residual` AFTER the activation          */
#define TRACE_EPI_RES_F32 32 /* `residual` is f32, not bf16 (trace_gemm_bf16_nt_ex) */
#define TRACE_EPI_POOL_T2 128 /* trace_conv3d_igemm_bf16, 1x1x1 stride-1 convs, even To: the epilogue also applies the
                               * MaxPool3d((2,1,1), stride (2,1,1)) that follows layer1 (resnet3d.py:270, :798): out is
                               * (N,To/2,Ho,Wo,Cout); residual keeps the unpooled shape.  Identical to conv + pool. */
#define TRACE_EPI_WTILED 64  /* trace_gemm_bf16_nt*: Bm is PRE-TILED for weight streaming (N % 128 == 0, K % 64 == 0,
                              * ldb == K): block (n/128, k/64) is 16 KiB contiguous, holding for row r = n % 128 and
                              * 16-byte chunk pc = 0..7 the 8 elements Bm[n][64*(k/64) + 8*(pc ^ (r & 7)) ..] at byte
                              * offset r*128 + pc*16 (the kernel's LDS image), blocks ordered [n/128][k/64].  A K-step
                              * then reads one contiguous block instead of 128 rows that lie K*2 bytes apart. */

/* Conv3d + folded eval-mode BatchNorm3d (+residual)(+ReLU) as an implicit GEMM
 * on bf16 MFMA (mmcv ConvModule at resnet3d.py:250-283,549-557,721-730;
 * residual add + ReLU resnet3d.py:306,313).
 *   in   (N,Ti,Hi,Wi,Cin)  bf16 channels-last, Cin % 8 == 0
 *   w    (Cout, kt*kh*kw*Cin) bf16, tap-major / channel-minor, BN scale folded
 *   bias (Cout) f32 (BN shift, or conv bias)
 *   out  (N,To,Ho,Wo,Cout) bf16 (or f32 with TRACE_EPI_OUT_F32), Cout % 8 == 0
 *   workspace: optional caller-owned device scratch (16-byte aligned; NULL = none).  When the
 *   tile grid cannot fill the chip and K is deep, K is split over workgroups that write f32
 *   partial slabs there and a second launch reduces them in slice order (deterministic). */
int trace_conv3d_igemm_bf16(const uint16_t* in, int N, int Ti, int Hi, int Wi,
                            int Cin, const uint16_t* w, const float* bias,
                            int Cout, int kt, int kh, int kw, int st, int sh,
                            int sw, int pt, int ph, int pw, int dt, int dh,
                            int dw, const uint16_t* residual, int epilogue,
                            void* out, void* workspace, size_t workspace_bytes,
                            trace_stream_t stream);

/* The I3D stem (resnet3d.py:721-730: Conv3d 3->64, k(5,7,7), s(2,2,2)) on the
 * halo-padded channels-last clip written by trace_clip_to_ndhwc_bf16: Cin=3 is
 * far too thin for an MFMA K-step, so one (kt,kh) tap is taken as the contiguous
 * run of kwp pixels x Cp channels (kwp*Cp % 8 == 0; weights packed
 * (Cout, kt*kh*kwp*Cp) with zeros in the padding slots).  `st` may fold the
 * temporal stride of the following MaxPool3d(kernel_t=1, stride_t=2)
 * (resnet3d.py:732-735), whose skipped frames are never computed. */
int trace_conv3d_stem_bf16(const uint16_t* in_padded, int N, int Tp, int Hp,
                           int Wp, int Cp, const uint16_t* w,
                           const float* bias, int Cout, int kt, int kh,
                           int kwp, int st, int sh, int sw, int To, int Ho,
                           int Wo, int epilogue, void* out,
                           trace_stream_t stream);

/* The same stem for TRACE's shape only (3 -> 64, k(5,7,7), pad (2,3,3), stride (st,2,2), Ho and Wo multiples of 16)
 * reading the caller's f32 clip directly - (N,C,T,H,W), or (N,T,C,H,W) with in_ntchw, C <= 3 - instead of a padded
 * channels-last bf16 copy: the kernel converts while it fills its LDS patch.  w is the (64, 5*7*8*4) bf16 matrix of
 * trace_conv3d_stem_bf16 (kwp = 8, Cp = 4).  Output (N,To,Ho,Wo,64) bf16, identical to the two-launch path. */
int trace_conv3d_stem16_f32(const float* clip, int N, int C, int T, int H, int W, int in_ntchw, const uint16_t* w,
                            const float* bias, int st, int To, int Ho, int Wo, int epilogue, void* out,
                            trace_stream_t stream);

/* The same stem from uint8 frames (N,T,H,W,C) channels-last, C <= 3, as a video decoder / cv2 leaves them:
 * value = (float)u8 - mean[c] (`mean`: C floats in HOST memory - cfg.PIXEL_MEANS - or NULL for 0), rounded to bf16
 * exactly as trace_conv3d_stem16_f32 rounds the f32 value, so the two entries give bit-identical outputs for the
 * same pixels.  Replaces the f32 blob of get_frames_blob (lib/core/test_engine_rel.py:420-520: mean-subtracted f32
 * frames, 4 bytes per sample over PCIe) when the frames are already at network resolution. */
int trace_conv3d_stem16_u8(const uint8_t* frames, int N, int T, int H, int W, int C, const float* mean,
                           const uint16_t* w, const float* bias, int st, int To, int Ho, int Wo, int epilogue,
                           void* out, trace_stream_t stream);

/* nn.MaxPool3d on channels-last bf16 (resnet3d.py:732-737; mmcv NonLocal3d
 * sub_sample pool).  Padding is implicit -inf as in torch. */
int trace_maxpool3d_ndhwc_bf16(const uint16_t* in, int N, int Ti, int Hi,
                               int Wi, int C, int kt, int kh, int kw, int st,
                               int sh, int sw, int pt, int ph, int pw,
                               uint16_t* out, trace_stream_t stream);

/* Batched bf16 GEMM  C[b] = alpha * A[b] (M,K) * B[b]^T (N,K)  (+bias)(+res)
 * (+ReLU); row strides lda/ldb/ldc in elements, batch strides in elements.
 * Used for the non-local dot-product attention (mmcv NonLocal3d.dot_product:
 * pairwise = theta^T phi / N_kv ; y = pairwise g) and every Linear layer of the
 * relation head.  K % 8 == 0, lda/ldb % 8 == 0. */
int trace_gemm_bf16_nt(const uint16_t* A, const uint16_t* Bm, int M, int N,
                       int K, int lda, int ldb, int ldc, int batch,
                       long long strideA, long long strideB, long long strideC,
                       float alpha, const float* bias,
                       const uint16_t* residual, int epilogue, void* C,
                       void* workspace, size_t workspace_bytes,
                       trace_stream_t stream);

/* Workspace sizing for the conv / GEMM family (SURVEY.md 8b "trace_workspace_bytes_*"; the reference's kernels allocate
 * their own scratch through ATen, e.g. Detectron_pytorch/lib/model/csrc/cuda/ROIAlign_cuda.cu:271 - here the caller
 * owns it).  Returns the split-K scratch the launch rules WANT for the problem: with at least that many bytes the plan
 * (tile, number of K slices and therefore the f32 summation order of the result) does not depend on the size of the
 * caller's buffer; with a smaller buffer fewer slices are used, with none the K loop is not split.  0 = no scratch
 * used.  Pure host functions: no device call, no launch. */
size_t trace_gemm_workspace_bytes(int M, int N, int K, int lda, int ldb, int ldc, int batch, int epilogue);
size_t trace_conv3d_workspace_bytes(int N, int Ti, int Hi, int Wi, int Cin, int Cout, int kt, int kh, int kw,
                                    int st, int sh, int sw, int pt, int ph, int pw, int dt, int dh, int dw,
                                    int epilogue);

/* trace_gemm_bf16_nt with the epilogue forms the relation head chains between its Linear layers
 * (model_builder_rel.py:938-952, 1018-1024, 1039-1069; cluster_tree3d.py:192-201, 293-330), so that a
 * sum or activation that the reference computes as a separate torch op costs no launch of its own:
 *   residual      bf16, or f32 with TRACE_EPI_RES_F32 (e.g. det_feat = ori + Linear(...), :946)
 *   strideBias    batch stride of `bias` (0 = shared)
 *   C2, epilogue2 optional second output, same shape / ldc / strideC as C: bf16(act(C value)) with
 *                 act = none | TRACE_EPI_RELU | TRACE_EPI_LEAKY - the operand of the next Linear when the
 *                 f32 value itself is also needed (e.g. ori and relu(ori), :938-939).  NULL = none. */
int trace_gemm_bf16_nt_ex(const uint16_t* A, const uint16_t* Bm, int M, int N, int K, int lda,
                          int ldb, int ldc, int batch, long long strideA, long long strideB,
                          long long strideC, float alpha, const float* bias, long long strideBias,
                          const void* residual, int epilogue, void* C, uint16_t* C2, int epilogue2,
                          void* workspace, size_t workspace_bytes, trace_stream_t stream);

/* C = act([A | A2] Bm^T + bias (+ residual)): one GEMM over two activation matrices with the same M rows - K-steps
 * below K1 read A (row stride lda), the rest A2 (lda2); Bm is (N, K1 + K2) row-major.  K1 and K2 multiples of 64.
 * A Bottleneck3d whose downsample conv has stride 1 (resnet3d.py:292-318: relu(conv3(h) + downsample(x))) becomes one
 * launch with Bm = [W3 | Wd]: the downsample output is neither written nor re-read as the residual. */
int trace_gemm_bf16_nt_dual(const uint16_t* A, int lda, int K1, const uint16_t* A2, int lda2, int K2, const uint16_t* Bm,
                            int ldb, int M, int N, int ldc, const float* bias, const void* residual, int epilogue, void* C,
                            trace_stream_t stream);
/* trace_gemm_bf16_nt_dual with the second output of trace_gemm_bf16_nt_ex: C2 (or NULL) = bf16(act2(C value)), same rows /
 * ldc.  With TRACE_EPI_OUT_F32 the block's f32 residual stream and the next conv's bf16 operand leave one launch. */
int trace_gemm_bf16_nt_dual_ex(const uint16_t* A, int lda, int K1, const uint16_t* A2, int lda2, int K2,
                               const uint16_t* Bm, int ldb, int M, int N, int ldc, const float* bias,
                               const void* residual, int epilogue, void* C, uint16_t* C2, int epilogue2,
                               trace_stream_t stream);


/* Launch-configuration override for the three entry points above, for the calling thread:
 * tile 1..9 = (64x64, 64 couts x 128 positions, 128 x 64, 128x128, 128 x 256, 256 x 128, 256x256, 128 x 192,
 * 256 x 192), 10 = the register-fed weight-streaming kernel, 12 = 128 x 160; nst = LDS ring depth 2..4,
 * ksplit = split-K factor; 0 = let the library choose (the default).  No counterpart in the
 * reference (cuDNN picks its own algorithm there); used by tools/tune_igemm.py and the tests that
 * cover every tile shape. */
int trace_igemm_tune(int tile, int nst, int ksplit);

/* bf16 (rows, cols) -> bf16 (cols, rows) per batch, output row stride ld_out
 * (>= rows) and batch stride out_batch_stride; used to form g^T for the
 * non-local block without leaving channels-last. */
int trace_transpose_bf16(const uint16_t* in, int batch, int rows, int cols,
                         uint16_t* out, int ld_out, long long out_batch_stride,
                         trace_stream_t stream);

/* Sub-sampled non-local block (mmcv NonLocal3d as configured by lib/i3d/
 * i3d_nl_dot_product_r50_32x2x1_100e_kinetics400_rgb.py:15-19; call site lib/i3d/resnet3d.py:289):
 * tpg (N,T,H,W,3c) bf16 = the theta|phi|g 1x1x1 convs run as one GEMM.  Max-pools (1,2,2) the phi and
 * g thirds (the `sub_sample` MaxPool3d) and writes phi_p (N,nkp,c) and g^T (N,c,nkp), the operand
 * layouts of the two products; nkp = T*(H/2)*(W/2) rounded up to 8, padding positions are zeros. */
int trace_nl_pool_split_bf16(const uint16_t* tpg, int N, int T, int H, int W, int c,
                             uint16_t* phi_p, uint16_t* gT, int nkp, trace_stream_t stream);
/* Same, for the reassociated non-local product y = theta (phi^T g) / N_kv (dot_product mode has no softmax, so
 * (theta phi^T) g = theta (phi^T g): the (N_q x N_kv) pairwise map is never formed): also / instead writes
 * phi^T (N, c, nkp) next to g^T.  phi_p or phiT may be NULL (not both). */
int trace_nl_pool_split2_bf16(const uint16_t* tpg, int N, int T, int H, int W, int c, uint16_t* phi_p,
                              uint16_t* phiT, uint16_t* gT, int nkp, trace_stream_t stream);

/* ------------------------------------------------------------------------ */
/* Context aggregation / relation head                                       */
/* ------------------------------------------------------------------------ */

/* cluster_tree3d.temporal_{obj,r}_attention — lib/combine_tree/
 * cluster_tree3d.py:337-366.  qs (n,D) f32 = W_s s ; qo (n,T,D) f32 = W_o o_t ;
 * o (n,T,K) f32 ; out (n,K) f32 = sum_t softmax_t(qs.qo_t/sqrt(D)) o_t. */
int trace_temporal_attention(const float* qs, const float* qo, const float* o,
                             int n, int T, int D, int K, float inv_sqrt_d,
                             float* out, trace_stream_t stream);
/* Same, with an optional bf16 copy of the result (the operand of the value MLP's first Linear,
 * cluster_tree3d.py:195); either output may be NULL. */
int trace_temporal_attention_ex(const float* qs, const float* qo, const float* o, int n, int T, int D,
                                int K, float inv_sqrt_d, float* out, uint16_t* out_bf16,
                                trace_stream_t stream);

/* Fused target-adaptive pair attention — lib/modeling_rel/
 * model_builder_rel.py:1524-1612 (get_spatio_atten_feat, get_rel_rois,
 * get_attention_mask_map): for every pair p, ROIAlign (adaptive grid) of the
 * K-map and V-map (B,H,W,D) bf16 on the union box of (sbj,obj), scores
 * q_s.K / q_o.K over the PH*PW cells, softmax, multiply into V.
 * Outputs s_map, o_map (rows of so_stride elements: D for two dense (P,PH*PW,D)
 * tensors, 2*D with o_map = s_map + D for the interleaved (P,PH*PW,2D) operand of
 * the following A|B 1x1-conv GEMM, which saves the torch.cat copy) and v_map
 * (P,PH*PW,D) bf16 — never materialising the pooled K-map in HBM. */
int trace_pair_attention_fused(const uint16_t* kmap, const uint16_t* vmap,
                               int B, int H, int W, int D,
                               const float* rel_rois, const int32_t* sbj_inds,
                               const int32_t* obj_inds, int P,
                               const float* q_s, const float* q_o,
                               float spatial_scale, int pooled, int so_stride,
                               uint16_t* s_map, uint16_t* o_map,
                               uint16_t* v_map, trace_stream_t stream);

/* Pair attention + spatio_inferA/B in one pass - model_builder_rel.py:1524-1612 and :1039-1041
 * (get_spatio_atten_feat, get_attention_mask_map, spatio_inferA / spatio_inferB, "+ V").  ROIAlign and the
 * 1x1 convs are linear over the map, so  A(s_map) + B(o_map) = a_s ROIAlign(W_A V) + a_o ROIAlign(W_B V) + b  and
 * q . ROIAlign(K) = ROIAlign(q . K):
 *   vab (B,H,W,3D) bf16 = [V | W_A V | W_B V] per position;  qk (2*n_obj, ld_qk >= B*H*W) f32, row 2i = q_s of
 *   object i against every position of every clip, row 2i+1 = q_o;  bias (D) f32 = b_A + b_B;
 *   sp (P, PH*PW, D) bf16 = LeakyReLU(a_s VAp + a_o VBp + bias) + Vp, a = softmax over the cells.
 * s_map / o_map / v_map are never written and the (P*PH*PW) x 2D x D GEMM of the reference's two convs is gone. */
/* partner (P) int32 or NULL: partner[p] = index of the pair with subject and object swapped, -1 if it is not in the
 * list.  rois_union (boxes_rel.py:100-106) is symmetric, so both pairs pool the same cells: the pair with the lower
 * index pools once and writes both rows of sp.  The caller guarantees rel_rois[partner[p]] == rel_rois[p] and
 * partner[partner[p]] == p.  Results are identical with and without it. */
int trace_pair_infer_fused(const uint16_t* vab, int B, int H, int W, int D, const float* qk, int ld_qk,
                           const float* rel_rois, const int32_t* sbj_inds, const int32_t* obj_inds,
                           const int32_t* partner, int P, const float* bias, float spatial_scale, int pooled,
                           uint16_t* sp, trace_stream_t stream);

/* trace_pair_infer_fused on the matrix cores (same references, same arguments where they share a name): the union-box
 * pooling as a matrix product  pooled[ch][cell] = sum_pix mapT[ch][pix] * Wt[pix][cell]  on v_mfma_f32_16x16x32_bf16,
 * scores, softmax over the cells and the combine in the epilogue, one workgroup per union box.
 *   vabT (B, 3D, ldp) bf16: [V | W_A V | W_B V] CHANNEL-major, pixel axis (y * W + x) contiguous with pitch ldp = 224,
 *     columns H*W .. ldp-1 zero, computed WITHOUT its bias;  vbias (3D) f32 (or NULL): that bias, added as
 *     vbias[ch] * (sum of the cell's weights);  maps of at most 16 x 16 pixels with H*W <= 200, pooled <= 8, D % 32 == 0.
 *   work (n_work x 12 int32, 16-byte aligned) or NULL: one record per union box to pool - [pair p, its swapped pair q or
 *     -1, sbj_inds[p], obj_inds[p], the box's five floats (bit patterns), 3 words of padding] - so that a workgroup reads
 *     everything it needs in one access and none is launched for the swapped pairs (with NULL: one workgroup per pair,
 *     inputs looked up through rel_rois / sbj_inds / obj_inds / partner as in trace_pair_infer_fused). */
int trace_pair_infer_mfma(const uint16_t* vabT, int B, int H, int W, int D, int ldp, const float* vbias,
                          const float* qk, int ld_qk, const float* rel_rois, const int32_t* sbj_inds,
                          const int32_t* obj_inds, const int32_t* partner, int P, const int32_t* work, int n_work,
                          const float* bias, float spatial_scale, int pooled, uint16_t* sp, trace_stream_t stream);

/* trace_pair_infer_fused as two launches (same references): trace_pair_pool writes the union-box pooling of
 * vab, out (P, PH*PW, 3D) bf16 = [Vp | (W_A V)p | (W_B V)p] - it needs only the 2-D map and the pair boxes, so a
 * caller can run it beside the I3D backbone; trace_pair_combine does what depends on the queries: pooled scores of
 * the scalar maps qk, softmax over the cells, sp = LeakyReLU(a_s VAp + a_o VBp + bias) + Vp. */
/* partner (or NULL) as in trace_pair_infer_fused: only the lower index of a swapped pair is pooled (row p of
 * `out`; the other row stays unwritten) and trace_pair_combine reads that row for both. */
int trace_pair_pool(const uint16_t* vab, int B, int H, int W, int D, const float* rel_rois, const int32_t* partner,
                    int P, float spatial_scale, int pooled, uint16_t* out, trace_stream_t stream);
int trace_pair_combine(const uint16_t* pooled_maps, int B, int H, int W, int D, const float* qk, int ld_qk,
                       const float* rel_rois, const int32_t* sbj_inds, const int32_t* obj_inds, const int32_t* partner,
                       int P, const float* bias, float spatial_scale, int pooled, uint16_t* sp, trace_stream_t stream);

/* Tree-GRU message passing — lib/combine_tree/cluster_tree.py:163-204
 * (forward_v1), 281-316 (bottom-up), 318-330 (top-down, 'tree' mode), for a
 * forest of per-clip box trees and all feature groups in one launch.  The
 * reference's dense adjacency masks (cluster_tree.py:206-215, uploaded 8x per
 * clip) are replaced by parent / CSR child lists built once.
 *   x (n_nodes, n_groups*F) f32 — group g reads columns [g*F,(g+1)*F)
 *   tree t owns nodes [tree_off[t], tree_off[t+1]); inside a tree nodes are
 *   numbered leaves first, then layer by layer (the reference numbering), and
 *   level l of tree t is the node range
 *   [level_off[t*(TRACE_TREE_MAX_LEVELS+1)+l], level_off[...+l+1]) (global ids)
 *   parent (n_nodes) global id or -1; children of node v are
 *   child_idx[child_off[v] .. child_off[v+1])
 *   h, hd (n_nodes, n_groups*Hd) f32 outputs (bottom-up / top-down states).
 * Weight matrices are passed TRANSPOSED, i.e. (in_features, Hd) row-major.
 * When max_tree_nodes bounds every tree and one tree's state plus one direction's
 * matrices fit in LDS, the whole pass runs out of LDS (a tree larger than the bound
 * gets NaN outputs instead of an LDS overrun). */
#define TRACE_TREE_MAX_LEVELS 12
#define TRACE_TREE_MAX_GROUPS 8
typedef struct trace_tree_gru_weights {
  /* bottom-up: Wz,Wr,Wh (F,Hd); Uz,Ur,Uh (Hd,Hd) [transposed]; bz,br (Hd) = Uz/Ur bias */
  const float *Wz, *Wr, *Wh, *Uz, *Ur, *Uh, *bz, *br;
  /* top-down: Wzd.. as above */
  const float *Wzd, *Wrd, *Whd, *Uzd, *Urd, *Uhd, *bzd, *brd;
} trace_tree_gru_weights;

int trace_tree_gru(const float* x, int n_nodes, int F, int Hd, int n_groups,
                   int n_trees, const int32_t* tree_off,
                   const int32_t* level_off, const int32_t* n_levels,
                   const int32_t* parent, const int32_t* child_off,
                   const int32_t* child_idx,
                   const trace_tree_gru_weights* weights /* host, n_groups */,
                   int max_tree_nodes /* upper bound of nodes in one tree; 0 = unknown */,
                   float* h, float* hd,
                   float* workspace /* n_nodes*n_groups*Hd f32 */,
                   trace_stream_t stream);
/* The same pass with the workspace size stated.  With trace_tree_gru_workspace_bytes(...) bytes (8 planes of
 * n_nodes*n_groups*Hd f32 when the wide kernels apply, else 1) matrices that do not fit LDS (Action Genome:
 * F = Hd = 256, model_builder_rel.py:1009-1016 with cluster_tree3d.py:70-75) run as batched products streamed from
 * L2 (tree_xterms_kernel + tree_gru_wide_kernel) instead of per-lane dot products; a smaller workspace (at least
 * one plane, else an error) selects the kernels of trace_tree_gru. */
size_t trace_tree_gru_workspace_bytes(int n_nodes, int F, int Hd, int n_groups);
int trace_tree_gru_ex(const float* x, int n_nodes, int F, int Hd, int n_groups,
                      int n_trees, const int32_t* tree_off,
                      const int32_t* level_off, const int32_t* n_levels,
                      const int32_t* parent, const int32_t* child_off,
                      const int32_t* child_idx,
                      const trace_tree_gru_weights* weights /* host, n_groups */,
                      int max_tree_nodes, float* h, float* hd,
                      float* workspace, size_t workspace_bytes, trace_stream_t stream);

/* Glue of the relation head (csrc/glue_ops.hip): each replaces a chain of small torch ops of the reference
 * by one launch.  `act` = 0 | TRACE_EPI_RELU | TRACE_EPI_LEAKY.
 *   act_cast        out (rows, wa+wb) bf16 = act([a | b]) (b optional, wb = 0): F.relu / LeakyReLU(0.1) in front
 *                   of a Linear (model_builder_rel.py:273,281-286,:939) and cat(h, hd) (cluster_tree.py:203)
 *   layernorm_act   nn.LayerNorm + ReLU of the temporal value MLPs (cluster_tree3d.py:80-84, 91-95)
 *   gather_add_act  out (rows,w) bf16 = act(a[r] + b[idx[r]]): low_s[X]+low_o[Y] + low_r[lca], LeakyReLU
 *                   (model_builder_rel.py:1662, :326)
 *   pair_assemble   per pair p = (X[p], Y[p]): labels / scores of subject and object (relpn_heads.py:113-116),
 *                   so_bias = low_s[X]+low_o[Y] (P,wl), so_part = part_s[X]+part_o[Y] (+ freq[(ls-1)*n_obj+lo-1],
 *                   sparse_targets_rel.py:67-72; freq may be NULL) (P,wp), model_builder_rel.py:1064-1075 */
int trace_act_cast_bf16(const float* a, int wa, const float* b, int wb, long long rows, int act,
                        uint16_t* out, trace_stream_t stream);
int trace_layernorm_act_bf16(const float* x, int rows, int w, const float* gamma, const float* beta,
                             float eps, int act, uint16_t* out, trace_stream_t stream);
int trace_gather_add_act_bf16(const float* a, const float* b, const int64_t* idx, long long rows, int w,
                              int act, uint16_t* out, trace_stream_t stream);
int trace_pair_assemble(const int64_t* det_labels, const float* det_scores, const int64_t* X,
                        const int64_t* Y, int P, const float* low_s, const float* low_o, int wl,
                        const float* part_s, const float* part_o, int wp, const float* freq, int n_rel,
                        int n_obj, int64_t* sbj_lab, int64_t* obj_lab, float* sbj_sc, float* obj_sc,
                        float* so_bias, float* so_part, trace_stream_t stream);
/* Same with optional per-pair labels / scores (each NULL or P entries): pair_sbj_lab / pair_obj_lab replace
 * det_labels[X] / det_labels[Y] (outputs and frequency-bias row), pair_sbj_sc / pair_obj_sc replace det_scores[X] /
 * det_scores[Y] - the `use_gt_labels` path of the gt-box branch (model_builder_rel.py:844-858, 871-876, 1051-1064),
 * where the labels come with the annotated pairs rather than with the de-duplicated boxes. */
int trace_pair_assemble_ex(const int64_t* det_labels, const float* det_scores, const int64_t* X, const int64_t* Y,
                           int P, const int64_t* pair_sbj_lab, const int64_t* pair_obj_lab, const float* pair_sbj_sc,
                           const float* pair_obj_sc, const float* low_s, const float* low_o, int wl, const float* part_s,
                           const float* part_o, int wp, const float* freq, int n_rel, int n_obj, int64_t* sbj_lab,
                           int64_t* obj_lab, float* sbj_sc, float* obj_sc, float* so_bias, float* so_part,
                           trace_stream_t stream);

/* Top-k triplet ranking - lib/datasets_rel/task_evaluation_vg_and_vrd.py:207-228 (+ argsort_desc,
 * lib/datasets_rel/pytorch_misc.py:28-35), for every clip of a step at once, so that only topk records per
 * clip leave the GPU.  prd (P_total, n_prd) f32 predicate scores; sbj/obj scores (P_total) f32 and labels
 * (P_total) int64 of the pairs; pair_off (n_clips+1) int64 DEVICE array, clip b owns pairs
 * [pair_off[b], pair_off[b+1]); max_pairs_per_clip is the host-known bound of a clip's pair count.
 * Per pair the prd_k best predicates, score = sbj*obj*prd, per clip the topk best cells in descending score
 * (ties: ascending (pair, slot) index = a stable descending sort).  ld_prd = row stride of prd (>= n_prd).  Outputs (n_clips, topk): scores f32,
 * clip-local pair index, predicate index, subject / object label (int64); rows of a clip with fewer than
 * topk candidates are padded with score -1.  A clip with up to 16384 candidates (pairs x min(prd_k,n_prd)) is sorted
 * in one pass on chip; a larger one (configs "100 objects, ~10k pairs") in segments of 16384 whose topk survivors are
 * merged by a second on-chip sort on the same (score, candidate index) order - the same list, ties included.
 * Limits: n_prd <= 256, segments x topk <= 16384 (error otherwise).
 * workspace, 256-aligned: trace_rank_triplets_workspace_bytes(P_total, min(prd_k,n_prd)) bytes for the one-pass case,
 * trace_rank_triplets_workspace_bytes_ex(...) for either case. */
size_t trace_rank_triplets_workspace_bytes(int P_total, int prd_k);
size_t trace_rank_triplets_workspace_bytes_ex(int P_total, int prd_k, int n_clips, int max_pairs_per_clip, int topk);
int trace_rank_triplets(const float* prd, int P_total, int n_prd, int ld_prd, const float* sbj_scores,
                        const float* obj_scores, const int64_t* sbj_labels, const int64_t* obj_labels,
                        const int64_t* pair_off, int n_clips, int max_pairs_per_clip, int prd_k, int topk,
                        float* top_scores, int64_t* top_pair, int64_t* top_prd, int64_t* top_sbj,
                        int64_t* top_obj, void* workspace, size_t workspace_bytes, trace_stream_t stream);

/* Per-step host arrays (roi lists, forest CSR arrays, pair indices) read straight from PINNED host memory by
 * the first kernel of a captured step - replaces the reference's blocking per-list uploads
 * (lib/modeling_rel/model_builder_rel.py:1389, lib/combine_tree/cluster_tree.py:214,730: torch.from_numpy(..).cuda()).
 * `pinned` holds two slots of slot_bytes; launch number k (kept in the device word counter[0], bumped by the
 * kernel; counter[1] is its arrival count: `counter` points at TWO zero-initialised words) copies nbytes of slot k & 1 to dst, so the host fills the other slot while the previous step runs.
 * Sizes multiples of 16, pointers 16-byte aligned; `pinned` must be device-accessible host memory. */
int trace_upload_slots(const void* pinned, size_t slot_bytes, unsigned* counter, void* dst, size_t nbytes,
                       trace_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TRACE_AMD_H */
