"""Mirror of the reference's `model._C` extension surface
(Detectron_pytorch/lib/model/csrc/vision.cpp:7-13) on top of libtrace_amd.so.

Same function names, argument order and error behaviour (RuntimeError) as the
pybind module; tensors live on the GPU; the ROI ops take float32 or float64 (the reference's dtype dispatch), NMS float32.
"""
import torch

from . import _lib as L


def _f32(t, name):
    L.require_gpu(t)
    if t.dtype != torch.float32:
        raise RuntimeError("%s must be float32 (got %s)" % (name, t.dtype))
    return t.contiguous()  # ROIAlign_cuda.cu:286,294


def _floating(t, name, like=None):
    """The reference dispatches these ops on float / double (AT_DISPATCH_FLOATING_TYPES) and reads every tensor of a call
    in the input's dtype: float32 or float64 here, all arguments alike."""
    L.require_gpu(t)
    if t.dtype not in (torch.float32, torch.float64):
        raise RuntimeError("%s must be float32 or float64 (got %s)" % (name, t.dtype))
    if like is not None and t.dtype != like.dtype:
        raise RuntimeError("%s is %s, the call's data are %s" % (name, t.dtype, like.dtype))
    return t.contiguous()  # ROIAlign_cuda.cu:286,294


def _sfx(t):
    return "f64" if t.dtype == torch.float64 else "f32"


def roi_align_forward(input, rois, spatial_scale, pooled_height, pooled_width, sampling_ratio):
    """vision.cpp:9 / ROIAlign.h:11-25 — returns (K,C,PH,PW)."""
    input = _floating(input, "input")
    rois = _floating(rois, "rois", input)
    if rois.dim() != 2 or rois.size(1) != 5:
        raise RuntimeError("rois must be (K,5)")
    B, C, H, W = input.shape
    K = rois.size(0)
    out = torch.empty((K, C, pooled_height, pooled_width), dtype=input.dtype, device=input.device)
    if out.numel() == 0:
        return out
    with torch.cuda.device(input.device):
        fn = getattr(L.lib(), "trace_roi_align_fwd_" + _sfx(input))
        L.check(fn(L.ptr(input), B, C, H, W, L.ptr(rois), K, float(spatial_scale), int(pooled_height),
                   int(pooled_width), int(sampling_ratio), L.ptr(out), L.stream()), "roi_align_forward")
    return out


def roi_align_backward(grad, rois, spatial_scale, pooled_height, pooled_width, batch_size, channels,
                       height, width, sampling_ratio):
    """vision.cpp:10 / ROIAlign.h:27-45 — returns (B,C,H,W)."""
    grad = _floating(grad, "grad")
    rois = _floating(rois, "rois", grad)
    gin = torch.empty((batch_size, channels, height, width), dtype=grad.dtype, device=grad.device)
    with torch.cuda.device(grad.device):
        fn = getattr(L.lib(), "trace_roi_align_bwd_" + _sfx(grad))
        L.check(fn(L.ptr(grad), L.ptr(rois), rois.size(0), float(spatial_scale), int(pooled_height),
                   int(pooled_width), int(batch_size), int(channels), int(height), int(width),
                   int(sampling_ratio), L.ptr(gin), L.stream()), "roi_align_backward")
    return gin


def roi_pool_forward(input, rois, spatial_scale, pooled_height, pooled_width):
    """vision.cpp:11 / ROIPool.h:11-25 — returns (output, argmax int32)."""
    input = _floating(input, "input")
    rois = _floating(rois, "rois", input)
    B, C, H, W = input.shape
    K = rois.size(0)
    out = torch.empty((K, C, pooled_height, pooled_width), dtype=input.dtype, device=input.device)
    argmax = torch.zeros((K, C, pooled_height, pooled_width), dtype=torch.int32, device=input.device)
    if out.numel() == 0:
        return out, argmax
    with torch.cuda.device(input.device):
        fn = getattr(L.lib(), "trace_roi_pool_fwd_" + _sfx(input))
        L.check(fn(L.ptr(input), B, C, H, W, L.ptr(rois), K, float(spatial_scale), int(pooled_height),
                   int(pooled_width), L.ptr(out), L.ptr(argmax), L.stream()), "roi_pool_forward")
    return out, argmax


def roi_pool_backward(grad, input, rois, argmax, spatial_scale, pooled_height, pooled_width,
                      batch_size, channels, height, width):
    """vision.cpp:12 / ROIPool.h:27-46 — returns (B,C,H,W)."""
    grad = _floating(grad, "grad")
    rois = _floating(rois, "rois", grad)
    L.require_gpu(argmax)
    argmax = argmax.contiguous()
    gin = torch.empty((batch_size, channels, height, width), dtype=grad.dtype, device=grad.device)
    with torch.cuda.device(grad.device):
        fn = getattr(L.lib(), "trace_roi_pool_bwd_" + _sfx(grad))
        L.check(fn(L.ptr(grad), L.ptr(argmax), L.ptr(rois), rois.size(0), int(pooled_height), int(pooled_width),
                   int(batch_size), int(channels), int(height), int(width), L.ptr(gin), L.stream()),
                "roi_pool_backward")
    return gin


NMS_GE = 0  # Cython / CPU reference semantics (cython_nms.pyx:84, nms_cpu.cpp:60)
NMS_GT = 1  # CUDA reference semantics (cuda/nms.cu:60)


NMS_MAX_BOXES = 16384  # one launch of trace_nms (csrc/nms.hip NMS_MAX_BOXES)


def nms(dets, scores, threshold, mode=NMS_GT):
    """vision.cpp:8 / nms.h:10-28 — dets (N,4), scores (N,) -> int64 keep indices.

    Returns kept original indices in ascending order on the dets' device (empty
    input -> empty CPU tensor as nms.h:17-18).  This binding only takes device
    tensors, which the reference's `_C.nms` dispatches to nms_cuda: suppression on
    IoU > thresh (cuda/nms.cu:60) - hence the default `mode`.  NMS_GE is the
    Cython / CPU rule (>=, cython_nms.pyx:84, nms_cpu.cpp:60) that TRACE's run-time
    NMS uses; `boxes.nms` / `boxes.nms_grouped` pass it explicitly.
    More than NMS_MAX_BOXES (16384) boxes raise: the reduce kernel keeps its "removed" bitmap (N / 64 words) and four mask
    words per lane in one workgroup.  TRACE's own calls stay below it (RPN: 6000 boxes before NMS; per-class NMS over the
    score-thresholded candidates of <= 1000 proposals); cut the candidate list by score first for anything larger.
    """
    if dets.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device="cpu")
    dets = _f32(dets, "dets")
    scores = _f32(scores, "scores")
    if dets.dim() != 2 or dets.size(1) != 4 or scores.numel() != dets.size(0):
        raise RuntimeError("nms expects dets (N,4) and scores (N,)")
    N = dets.size(0)
    lib = L.lib()
    ws_bytes = lib.trace_nms_workspace_bytes(N)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dets.device)
    keep = torch.empty((N,), dtype=torch.int64, device=dets.device)
    num = torch.zeros((1,), dtype=torch.int32, device=dets.device)
    with torch.cuda.device(dets.device):
        L.check(lib.trace_nms(L.ptr(dets), L.ptr(scores), N, float(threshold), int(mode), L.ptr(keep),
                              L.ptr(num), L.ptr(ws), ws_bytes, L.stream()), "nms")
    return keep[: int(num.item())]
