"""TEST INFRASTRUCTURE ONLY.  Compile the reference's own CPU `model._C`
sources (vision.cpp:7-13, cpu/ROIAlign_cpu.cpp, cpu/nms_cpu.cpp) from where they
lie under /root/reference into oracle/_ref/trace_ref_C.so.  Nothing is copied
into the repo; oracle/_ref/ is git-ignored but travels to the GPU box.

Only tests/, bench.py's cpu_baseline leg and __graft_entry__ may use the result.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF_CSRC = "/root/reference/Detectron_pytorch/lib/model/csrc"
OUT = os.path.join(HERE, "_ref")
NAME = "trace_ref_C"


def build(verbose=False):
    if not os.path.isdir(REF_CSRC):
        return None
    os.makedirs(OUT, exist_ok=True)
    from torch.utils.cpp_extension import load
    mod = load(
        name=NAME,
        sources=[os.path.join(REF_CSRC, "vision.cpp"),
                 os.path.join(REF_CSRC, "cpu", "ROIAlign_cpu.cpp"),
                 os.path.join(REF_CSRC, "cpu", "nms_cpu.cpp")],
        extra_include_paths=[REF_CSRC],
        extra_cflags=["-O2", "-w", "-include", os.path.join(HERE, "ref_compat.h")],
        build_directory=OUT,
        verbose=verbose,
    )
    return mod


def load_ref():
    """Import the prebuilt oracle/_ref/trace_ref_C.so (no compilation)."""
    import importlib.util
    import torch  # noqa: F401  (libtorch must be loaded first)
    path = os.path.join(OUT, NAME + ".so")
    if not os.path.exists(path):
        return None
    spec = importlib.util.spec_from_file_location(NAME, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    m = build(verbose=True)
    print("built" if m is not None else "reference not present", file=sys.stderr)
