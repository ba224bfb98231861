"""Import environment for running the REFERENCE's own Python on the CPU of the build container
(only here: /root/reference does not travel).  Used by tools/make_goldens_py.py to produce
tests/golden/*.npz; nothing in the product, the tests or the bench imports this module.

What is adapted, and why none of it touches the arithmetic that the golden vectors pin:
  * sys.path gets the reference's own roots (tools/, lib/, Detectron_pytorch/lib), as its
    tools/_init_paths.py does;
  * EMPTY placeholder modules (names only, no behaviour) for packages the image lacks and whose
    symbols are imported at module top but never reached by the functions we call: torchvision,
    cv2, pycocotools, mmcv (ConvModule/NonLocal3d are set to None: anything that tried to use them
    would raise), the three Cython extensions (cython_bbox, cython_nms, cython_bbox_rel: set to None,
    so the NMS / overlap paths cannot be exercised through Python here - they are pinned through the
    compiled csrc/cpu/nms_cpu.cpp instead);
  * `torch._six` (removed from torch 2.x; the reference reads string_classes/int_classes from it) and
    the numpy aliases `np.int`, `np.float` removed in numpy 1.24 - version compatibility only;
  * `model._C` is the reference's own C++ (csrc/vision.cpp + cpu/*.cpp) compiled by oracle/build_ref.py;
  * `cfg.BBOX_XFORM_CLIP` is turned from np.float64 into the same value as a Python float (numpy 1.x casting
    semantics under numpy 2, see the comment in setup());
  * `torch.Tensor.cuda` returns the tensor itself, made contiguous (there is no GPU in this container and the
    reference hard-codes `.cuda(device_id)` after `torch.from_numpy`, cluster_tree.py:214,730,
    model_builder_rel.py:1389) - placement only, values unchanged.  Contiguous because the union-roi array is a
    transposed view (model_builder_rel.py:1594): the CUDA launcher the reference runs reads
    `rois.contiguous()` (ROIAlign_cuda.cu:286-294) while its CPU kernel reads `rois.data()` as is
    (ROIAlign_cpu.cpp:252), so without this the CPU kernel would see scrambled boxes the real path never sees.
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _empty(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def setup():
    """Returns the reference's `cfg` after making its modules importable."""
    if not os.path.isdir(REF):
        raise RuntimeError("reference tree not present: goldens can only be generated in the build container")
    if "core.config" in sys.modules:
        return sys.modules["core.config"].cfg
    tv = _empty("torchvision")
    for sub in ("models", "transforms", "ops"):
        setattr(tv, sub, _empty("torchvision." + sub))
    cv2 = _empty("cv2")
    cv2.ocl = types.SimpleNamespace(setUseOpenCL=lambda *a: None)
    cv2.setNumThreads = lambda *a: None
    pc = _empty("pycocotools")
    for sub in ("mask", "coco", "cocoeval"):
        setattr(pc, sub, _empty("pycocotools." + sub))
    sys.modules["pycocotools.coco"].COCO = None
    sys.modules["pycocotools.cocoeval"].COCOeval = None
    mm = _empty("mmcv")
    mm.cnn = _empty("mmcv.cnn", ConvModule=None, NonLocal3d=None, build_activation_layer=None,
                    constant_init=None, kaiming_init=None)
    mm.runner = _empty("mmcv.runner", _load_checkpoint=None, load_checkpoint=None)
    mm.utils = _empty("mmcv.utils", _BatchNorm=None, get_logger=None)
    _empty("utils.cython_bbox", bbox_overlaps=None)
    _empty("utils.cython_nms", nms=None, soft_nms=None)
    _empty("utils_rel.cython_bbox_rel", bbox_overlaps=None, bbox_pair_overlaps=None)
    import collections.abc
    torch._six = _empty("torch._six", string_classes=(str,), int_classes=(int,), container_abcs=collections.abc)
    if not hasattr(np, "int"):
        np.int = int
    if not hasattr(np, "float"):
        np.float = float
    sys.path[:0] = [REF + "/tools", REF + "/lib", REF + "/Detectron_pytorch/lib"]
    if ROOT not in sys.path:
        sys.path.append(ROOT)
    from oracle.build_ref import build, load_ref
    ref_c = load_ref() or build()
    assert ref_c is not None, "could not build the reference's CPU extension"
    import model as ref_model_pkg
    ref_model_pkg._C = ref_c
    sys.modules["model._C"] = ref_c
    torch.Tensor.cuda = lambda self, *a, **k: self.contiguous()
    from core.config import cfg
    # numpy >= 2 (NEP 50) treats the np.float64 scalar np.log(1000/16) as strongly typed, which would promote
    # bbox_transform's float32 dw/dh to float64; under the numpy 1.x the reference was written for (it uses
    # np.int) value-based casting keeps them float32.  A Python float restores exactly that behaviour.
    cfg.BBOX_XFORM_CLIP = float(cfg.BBOX_XFORM_CLIP)
    return cfg
