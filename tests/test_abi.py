"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports
every symbol include/trace_amd.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "trace_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(trace_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from trace_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build with python -m trace_amd.build"
    dll = ctypes.CDLL(_lib.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 20
    missing = [s for s in syms if not hasattr(dll, s)]
    assert not missing, "missing exports: %s" % missing


def test_binding_table_matches_header():
    from trace_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()


def test_abi_version_and_error_string():
    from trace_amd import _lib
    l = _lib.lib()
    assert l.trace_abi_version() == 1
    assert isinstance(l.trace_last_error(), bytes)


def test_cpu_tensors_are_rejected_loudly():
    import torch
    from trace_amd import roi_layers
    with pytest.raises(RuntimeError):
        roi_layers.ROIAlign((7, 7), 1 / 16., 2)(torch.zeros(1, 4, 8, 8), torch.zeros(1, 5))
    with pytest.raises(RuntimeError):
        roi_layers.nms(torch.zeros(3, 4), torch.zeros(3), 0.5)


def test_empty_nms_returns_empty_cpu_tensor():
    # csrc/nms.h:17-18
    import torch
    from trace_amd import roi_layers
    out = roi_layers.nms(torch.zeros(0, 4), torch.zeros(0), 0.5)
    assert out.numel() == 0 and out.dtype == torch.int64 and out.device.type == "cpu"


def test_gemm_workspace_bytes_is_the_split_k_plan():
    """trace_gemm_workspace_bytes / trace_conv3d_workspace_bytes (host-only): the scratch the launch rules want - K slices
    x M x N f32 slabs for the deep-K weight-streaming FCs of the relation head, nothing for batched or shallow problems."""
    from trace_amd import _lib
    l = _lib.lib()
    EPI_WTILED = 64
    fc1 = l.trace_gemm_workspace_bytes(160, 3072, 100352, 100352, 100352, 3072, 1, EPI_WTILED)
    assert fc1 % (160 * 3072 * 4) == 0 and 8 <= fc1 // (160 * 3072 * 4) <= 32
    assert l.trace_gemm_workspace_bytes(6272, 1024, 256, 256, 256, 1024, 1, 0) == 0          # one round of K: no split
    assert l.trace_gemm_workspace_bytes(512, 512, 200, 200, 200, 512, 8, 0) == 0             # batched: never split
    one_clip = l.trace_conv3d_workspace_bytes(1, 1, 14, 14, 2048, 512, 1, 1, 1, 1, 1, 1, 0, 0, 0, 1, 1, 1, 1)
    assert one_clip % (196 * 512 * 4) == 0 and one_clip > 0
    assert l.trace_conv3d_workspace_bytes(8, 4, 14, 14, 512, 512, 1, 3, 3, 1, 1, 1, 0, 2, 2, 1, 2, 2, 1) == 0   # patch-shared 3x3
