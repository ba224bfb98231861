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
