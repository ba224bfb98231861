"""TEST INFRASTRUCTURE ONLY — numpy-facing wrappers of oracle/_build/liboracle.so
(the plain-C restatement in oracle/trace_oracle.c)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build():
    subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_nms.restype = ctypes.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _real(inp):
    """(numpy dtype, symbol suffix, scale ctype) of a ROI call: float64 data take the _f64 build of roi_oracle.inc (the
    reference passes spatial_scale as a C float whatever the tensor type: it is rounded to float32 first)."""
    if np.asarray(inp).dtype == np.float64:
        return np.float64, "_f64", lambda s: ctypes.c_double(float(np.float32(s)))
    return np.float32, "", lambda s: ctypes.c_float(s)


def roi_align_fwd(inp, rois, scale, ph, pw, sr):
    dt, sfx, sc = _real(inp)
    inp, rois = np.ascontiguousarray(inp, dt), np.ascontiguousarray(rois, dt)
    B, C, H, W = inp.shape
    K = rois.shape[0]
    out = np.empty((K, C, ph, pw), dt)
    if out.size:
        getattr(lib(), "oracle_roi_align_fwd" + sfx)(_p(inp), B, C, H, W, _p(rois), K, sc(scale), ph, pw, sr, _p(out))
    return out


def roi_align_bwd(grad, rois, scale, ph, pw, B, C, H, W, sr):
    dt, sfx, sc = _real(grad)
    grad, rois = np.ascontiguousarray(grad, dt), np.ascontiguousarray(rois, dt)
    gin = np.zeros((B, C, H, W), dt)
    if grad.size:
        getattr(lib(), "oracle_roi_align_bwd" + sfx)(_p(grad), _p(rois), rois.shape[0], sc(scale), ph, pw, B, C, H, W, sr, _p(gin))
    return gin


def roi_pool_fwd(inp, rois, scale, ph, pw):
    dt, sfx, sc = _real(inp)
    inp, rois = np.ascontiguousarray(inp, dt), np.ascontiguousarray(rois, dt)
    B, C, H, W = inp.shape
    K = rois.shape[0]
    out = np.empty((K, C, ph, pw), dt)
    am = np.zeros((K, C, ph, pw), np.int32)
    if out.size:
        getattr(lib(), "oracle_roi_pool_fwd" + sfx)(_p(inp), B, C, H, W, _p(rois), K, sc(scale), ph, pw, _p(out), _p(am))
    return out, am


def roi_pool_bwd(grad, argmax, rois, ph, pw, B, C, H, W):
    dt, sfx, _ = _real(grad)
    grad, rois = np.ascontiguousarray(grad, dt), np.ascontiguousarray(rois, dt)
    argmax = np.ascontiguousarray(argmax, np.int32)
    gin = np.zeros((B, C, H, W), dt)
    if grad.size:
        getattr(lib(), "oracle_roi_pool_bwd" + sfx)(_p(grad), _p(argmax), _p(rois), rois.shape[0], ph, pw, B, C, H, W, _p(gin))
    return gin


def nms(boxes, scores, thresh, mode=0):
    boxes, scores = _f(boxes), _f(scores)
    N = boxes.shape[0]
    keep = np.empty((max(N, 1),), np.int64)
    k = lib().oracle_nms(_p(boxes), _p(scores), N, ctypes.c_float(thresh), mode, _p(keep))
    return keep[:k].copy()


def bbox_overlaps(boxes, query):
    boxes, query = _f(boxes), _f(query)
    out = np.zeros((boxes.shape[0], query.shape[0]), np.float32)
    if out.size:
        lib().oracle_bbox_overlaps(_p(boxes), boxes.shape[0], _p(query), query.shape[0], _p(out))
    return out


def bbox_pair_overlaps(b1, b2):
    b1, b2 = _f(b1), _f(b2)
    out = np.zeros((b1.shape[0],), np.float32)
    if out.size:
        lib().oracle_bbox_pair_overlaps(_p(b1), _p(b2), b1.shape[0], _p(out))
    return out
