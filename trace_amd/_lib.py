"""ctypes binding of libtrace_amd.so (the C-ABI declared in include/trace_amd.h).

There is no CPU fallback: if the HIP library is missing or a tensor is not on a
GPU the call raises.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# TRACE_AMD_LIB: load another build of the same library (A/B diagnostics); the default is the in-tree build
LIB_PATH = os.environ.get("TRACE_AMD_LIB") or os.path.join(_HERE, "lib", "libtrace_amd.so")

c_int = ctypes.c_int
c_float = ctypes.c_float
c_void_p = ctypes.c_void_p
c_size_t = ctypes.c_size_t
c_longlong = ctypes.c_longlong


class TreeGruWeights(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in ("Wz", "Wr", "Wh", "Uz", "Ur", "Uh", "bz", "br",
                                        "Wzd", "Wrd", "Whd", "Uzd", "Urd", "Uhd", "bzd", "brd")]


P = c_void_p
# name -> (restype, argtypes); must list every symbol of include/trace_amd.h
SIGNATURES = {
    "trace_last_error": (ctypes.c_char_p, []),
    "trace_abi_version": (c_int, []),
    "trace_roi_align_fwd_f32": (c_int, [P, c_int, c_int, c_int, c_int, P, c_int, c_float, c_int, c_int, c_int, P, P]),
    "trace_roi_align_bwd_f32": (c_int, [P, P, c_int, c_float, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P]),
    "trace_roi_pool_fwd_f32": (c_int, [P, c_int, c_int, c_int, c_int, P, c_int, c_float, c_int, c_int, P, P, P]),
    "trace_roi_pool_bwd_f32": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P]),
    "trace_roi_align_fwd_f64": (c_int, [P, c_int, c_int, c_int, c_int, P, c_int, c_float, c_int, c_int, c_int, P, P]),
    "trace_roi_align_bwd_f64": (c_int, [P, P, c_int, c_float, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P]),
    "trace_roi_pool_fwd_f64": (c_int, [P, c_int, c_int, c_int, c_int, P, c_int, c_float, c_int, c_int, P, P, P]),
    "trace_roi_pool_bwd_f64": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P]),
    "trace_roi_align_fwd_nhwc_bf16": (c_int, [P, c_int, c_int, c_int, c_int, P, c_int, c_float, c_int, c_int, c_int, P, P]),
    "trace_roi_align_avg_nhwc_bf16": (c_int, [P, c_int, c_int, c_int, c_int, P, c_int, c_float, c_int, c_int, c_int, P, P]),
    "trace_nl_pool_split_bf16": (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, P, c_int, P]),
    "trace_nl_pool_split2_bf16": (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, P, P, c_int, P]),
    "trace_rank_triplets_workspace_bytes": (c_size_t, [c_int, c_int]),
    "trace_rank_triplets_workspace_bytes_ex": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "trace_rank_triplets": (c_int, [P, c_int, c_int, c_int, P, P, P, P, P, c_int, c_int, c_int, c_int, P, P, P, P, P, P, c_size_t, P]),
    "trace_act_cast_bf16": (c_int, [P, c_int, P, c_int, c_longlong, c_int, P, P]),
    "trace_layernorm_act_bf16": (c_int, [P, c_int, c_int, P, P, c_float, c_int, P, P]),
    "trace_gather_add_act_bf16": (c_int, [P, P, P, c_longlong, c_int, c_int, P, P]),
    "trace_pair_assemble": (c_int, [P, P, P, P, c_int, P, P, c_int, P, P, c_int, P, c_int, c_int, P, P, P, P, P, P, P]),
    "trace_pair_assemble_ex": (c_int, [P, P, P, P, c_int, P, P, P, P, P, P, c_int, P, P, c_int, P, c_int, c_int, P, P, P, P, P, P, P]),
    "trace_temporal_attention_ex": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_float, P, P, P]),
    "trace_nms_workspace_bytes": (c_size_t, [c_int]),
    "trace_nms": (c_int, [P, P, c_int, c_float, c_int, P, P, P, c_size_t, P]),
    "trace_nms_grouped": (c_int, [P, P, P, c_int, c_float, c_int, P, P, P, c_size_t, P]),
    "trace_nms_noalloc": (c_int, [P, P, c_int, c_float, c_int, P, P, P]),
    "trace_bbox_transform_clip": (c_int, [P, P, c_int, c_int] + [c_float] * 7 + [P, P]),
    "trace_fpn_levels": (c_int, [P, c_int, c_int, c_int, c_float, c_float, P, P]),
    "trace_bbox_overlaps": (c_int, [P, c_int, P, c_int, P, P]),
    "trace_bbox_pair_overlaps": (c_int, [P, P, c_int, P, P]),
    "trace_relpn_pairs": (c_int, [P, c_int, P, P, P, P]),
    "trace_gemm_workspace_bytes": (c_size_t, [c_int] * 8),
    "trace_conv3d_workspace_bytes": (c_size_t, [c_int] * 19),
    "trace_clip_to_ndhwc_bf16": (c_int, [P] + [c_int] * 13 + [P, P]),
    "trace_conv3d_stem_bf16": (c_int, [P] + [c_int] * 5 + [P, P] + [c_int] * 11 + [P, P]),
    "trace_conv3d_stem16_f32": (c_int, [P] + [c_int] * 6 + [P, P] + [c_int] * 5 + [P, P]),
    "trace_conv3d_stem16_u8": (c_int, [P] + [c_int] * 5 + [P, P, P] + [c_int] * 5 + [P, P]),
    "trace_conv3d_igemm_bf16": (c_int, [P] + [c_int] * 5 + [P, P] + [c_int] * 13 + [P, c_int, P, P, c_size_t, P]),
    "trace_maxpool3d_ndhwc_bf16": (c_int, [P] + [c_int] * 14 + [P, P]),
    "trace_gemm_bf16_nt_dual": (c_int, [P, c_int, c_int, P, c_int, c_int, P, c_int, c_int, c_int, c_int, P, P, c_int, P, P]),
    "trace_gemm_bf16_nt_dual_ex": (c_int, [P, c_int, c_int, P, c_int, c_int, P, c_int, c_int, c_int, c_int, P, P, c_int, P, P, c_int, P]),
    "trace_gemm_bf16_nt_ex": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_longlong, c_longlong, c_longlong, c_float, P, c_longlong, P, c_int, P, P, c_int, P, c_size_t, P]),
    "trace_gemm_bf16_nt": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_longlong, c_longlong, c_longlong, c_float, P, P, c_int, P, P, c_size_t, P]),
    "trace_igemm_tune": (c_int, [c_int, c_int, c_int]),
    "trace_transpose_bf16": (c_int, [P, c_int, c_int, c_int, P, c_int, c_longlong, P]),
    "trace_temporal_attention": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_float, P, P]),
    "trace_pair_attention_fused": (c_int, [P, P, c_int, c_int, c_int, c_int, P, P, P, c_int, P, P, c_float, c_int, c_int, P, P, P, P]),
    "trace_upload_slots": (c_int, [P, c_size_t, P, P, c_size_t, P]),
    "trace_pair_infer_fused": (c_int, [P, c_int, c_int, c_int, c_int, P, c_int, P, P, P, P, c_int, P, c_float, c_int, P, P]),
    "trace_pair_infer_mfma": (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, P, c_int, P, P, P, P, c_int, P, c_int, P, c_float, c_int, P, P]),
    "trace_pair_pool": (c_int, [P, c_int, c_int, c_int, c_int, P, P, c_int, c_float, c_int, P, P]),
    "trace_pair_combine": (c_int, [P, c_int, c_int, c_int, c_int, P, c_int, P, P, P, P, c_int, P, c_float, c_int, P, P]),
    "trace_tree_gru": (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, P,
                               ctypes.POINTER(TreeGruWeights), c_int, P, P, P, P]),
    "trace_tree_gru_ex": (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, P,
                                  ctypes.POINTER(TreeGruWeights), c_int, P, P, P, c_size_t, P]),
    "trace_tree_gru_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
}

_lib = None


def lib():
    """Load libtrace_amd.so (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "trace_amd: %s is missing - build it with `python -m trace_amd.build` "
                "(there is no CPU fallback)" % LIB_PATH)
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().trace_last_error()
        raise RuntimeError("trace_amd %s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return c_void_p(t.data_ptr())


def stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("trace_amd: tensor must live on a GPU (no CPU path in this library)")
