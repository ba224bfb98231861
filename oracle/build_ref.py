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


# ---- the reference's Cython overlap helpers (round 3: they do build with the image's Cython 3.2 - only cython_nms.pyx,
# which declares np.int_t buffers, does not; its algorithm is pinned through nms_cpu.cpp above)
REF_PYX = {"cython_bbox": "/root/reference/Detectron_pytorch/lib/utils/cython_bbox.pyx",          # bbox_overlaps :32-73
           "cython_bbox_rel": "/root/reference/lib/utils_rel/cython_bbox_rel.pyx"}                # bbox_pair_overlaps :37-77


def build_cython(verbose=False):
    """cythonize + gcc the two .pyx files from where they lie; the generated C goes to a temporary directory and is
    deleted, only the extension modules land in oracle/_ref/.  Returns the list of modules built."""
    import subprocess
    import sysconfig
    import tempfile
    import numpy
    if not all(os.path.exists(v) for v in REF_PYX.values()):
        return []
    os.makedirs(OUT, exist_ok=True)
    suf = sysconfig.get_config_var("EXT_SUFFIX")
    built = []
    with tempfile.TemporaryDirectory() as tmp:
        for name, pyx in REF_PYX.items():
            c = os.path.join(tmp, name + ".c")
            subprocess.check_call([sys.executable, "-m", "cython", "-3", pyx, "-o", c],
                                  stdout=None if verbose else subprocess.DEVNULL, stderr=None if verbose else subprocess.DEVNULL)
            subprocess.check_call(["gcc", "-O2", "-w", "-shared", "-fPIC", "-DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION",
                                   "-I" + sysconfig.get_paths()["include"], "-I" + numpy.get_include(), c, "-o",
                                   os.path.join(OUT, name + suf)])
            built.append(name)
    return built


def load_cython(name):
    """Import oracle/_ref/<name>.<abi>.so built by build_cython (None when it is not there)."""
    import importlib.util
    import sysconfig
    path = os.path.join(OUT, name + sysconfig.get_config_var("EXT_SUFFIX"))
    if not os.path.exists(path):
        return None
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    m = build(verbose=True)
    print("built" if m is not None else "reference not present", file=sys.stderr)
    print("cython modules:", build_cython(verbose=True), file=sys.stderr)
