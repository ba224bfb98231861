"""Round-2 fixtures, again produced by RUNNING THE REFERENCE'S OWN PYTHON in the build container (environment:
tools/ref_env.py); data only, no reference text is stored.

    python tools/make_goldens_r2.py

  py_gt_branch.npz    the gt-box branch's host arithmetic, lib/modeling_rel/model_builder_rel.py:773-810 and :845-858.
                      These lines are inline in `_forward` (no function boundary), so the generator reads exactly those
                      line ranges from the reference file AT RUN TIME, dedents them and executes them on synthetic roidb
                      records (duplicate pairs, shared boxes, one pair only); inputs and resulting arrays are stored.
  py_i3d_wiring.npz   the I3D-R50 wiring: lib/i3d/resnet3d.py `ResNet3d(**backbone)` is constructed with the DC5
                      overrides of model_builder_rel.py:166-183 applied to lib/i3d/i3d_nl_dot_product_...py's backbone
                      dict, with RECORDING stand-ins for mmcv's ConvModule / NonLocal3d (constructor arguments only, no
                      arithmetic).  Stored: per module name the constructor arguments, in construction order.
"""
import json
import os
import sys
import textwrap
import types

import numpy as np
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_env  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
REF_FILE = os.path.join(ref_env.REF, "lib/modeling_rel/model_builder_rel.py")


def jbytes(obj):
    return np.frombuffer(json.dumps(obj).encode(), dtype=np.uint8)


def ref_lines(first, last):
    with open(REF_FILE) as f:
        lines = f.readlines()[first - 1:last]
    return textwrap.dedent("".join(lines))


def gt_branch(cfg):
    import utils.blob as blob_utils
    import utils_rel.boxes_rel as box_utils_rel
    import torch
    part1 = compile(ref_lines(773, 810), "model_builder_rel.py:773-810", "exec")
    part2 = compile(ref_lines(845, 858), "model_builder_rel.py:845-858", "exec")
    rng = np.random.default_rng(5)
    g = {}
    cases = []

    def boxes(n):
        x1 = rng.integers(0, 300, n)
        y1 = rng.integers(0, 200, n)
        return np.stack([x1, y1, x1 + rng.integers(8, 200, n), y1 + rng.integers(8, 150, n)], 1).astype(np.float32)

    for name, n_obj, n_pair in (("p1", 2, 1), ("p9", 5, 9), ("p40", 12, 40), ("p6_dup", 4, 6)):
        bx = boxes(n_obj)
        cls = rng.integers(0, 35, n_obj).astype(np.int32)
        s = rng.integers(0, n_obj, n_pair)
        o = (s + rng.integers(1, n_obj, n_pair)) % n_obj
        if name.endswith("dup"):                  # the same annotated pair several times (multi-label relations)
            s[3:], o[3:] = s[:3], o[:3]
            bx[3] = bx[0]                         # and two objects with identical boxes
            cls[3] = (cls[0] + 1) % 35
        roidb = dict(sbj_gt_boxes=bx[s].copy(), obj_gt_boxes=bx[o].copy(), sbj_gt_classes=cls[s].copy(),
                     obj_gt_classes=cls[o].copy())
        im_info = torch.tensor([[600.0, 800.0, 1.6]])
        ns = dict(np=np, roidb=roidb, im_info=im_info, blob_utils=blob_utils, box_utils_rel=box_utils_rel)
        exec(part1, ns)
        exec(part2, ns)
        cases.append(name)
        for k in ("sbj_gt_boxes", "obj_gt_boxes", "sbj_gt_classes", "obj_gt_classes"):
            g[name + "_in_" + k] = roidb[k]
        g[name + "_in_im_info"] = im_info.numpy()
        rel_ret = ns["rel_ret"]
        for k in ("sbj_rois", "obj_rois", "rel_rois", "det_rois", "sbj_inds", "obj_inds"):
            g[name + "_out_" + k] = np.asarray(rel_ret[k])
        for k in ("pair_gt_index", "sbj_labels", "obj_labels", "sbj_scores", "obj_scores", "det_labels", "det_scores"):
            g[name + "_out_" + k] = np.asarray(ns[k])
    g["cases"] = jbytes(cases)
    np.savez_compressed(os.path.join(OUT, "py_gt_branch.npz"), **g)


RECORDS = []


class ConvModule(nn.Module):
    """Recording stand-in: keeps its constructor arguments, computes nothing."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias='auto',
                 conv_cfg=None, norm_cfg=None, act_cfg=dict(type='ReLU'), **kw):
        super().__init__()
        self.rec = dict(kind="ConvModule", in_channels=in_channels, out_channels=out_channels, kernel_size=kernel_size,
                        stride=stride, padding=padding, dilation=dilation, groups=groups, bias=bias,
                        conv_type=None if conv_cfg is None else conv_cfg.get("type"),
                        norm_type=None if norm_cfg is None else norm_cfg.get("type"),
                        act_type=None if act_cfg is None else act_cfg.get("type"))
        if norm_cfg is not None:
            self.norm = types.SimpleNamespace(num_features=out_channels)   # resnet3d.py:289 reads conv3.norm.num_features
        RECORDS.append(self.rec)


class NonLocal3d(nn.Module):
    def __init__(self, in_channels, **kw):
        super().__init__()
        kw = dict(kw)
        if "norm_cfg" in kw and kw["norm_cfg"] is not None:
            kw["norm_cfg"] = dict(kw["norm_cfg"])
        self.rec = dict(kind="NonLocal3d", in_channels=in_channels, kwargs=kw)
        RECORDS.append(self.rec)


def i3d_wiring(cfg):
    import importlib
    mm = sys.modules["mmcv.cnn"]
    mm.ConvModule, mm.NonLocal3d = ConvModule, NonLocal3d
    mm.build_activation_layer = lambda act_cfg: nn.Identity()       # placeholder module: nothing is ever run
    sys.modules["mmcv.utils"]._BatchNorm = nn.modules.batchnorm._BatchNorm
    r3d = importlib.import_module("i3d.resnet3d")
    i3d_cfg = importlib.import_module("i3d.i3d_nl_dot_product_r50_32x2x1_100e_kinetics400_rgb")
    backbone = dict(i3d_cfg.model['backbone'])
    # the DC5 overrides, model_builder_rel.py:166-183 (cfg.I3D_DC5 = True, FREEZE_PART_I3D = False)
    backbone['frozen_stages'] = 0
    backbone['num_stages'] = 4
    backbone['dilations'] = (1, 1, 1, 2)
    backbone['spatial_strides'] = (1, 2, 2, 1)
    backbone.pop('type', None)
    net = r3d.ResNet3d(**backbone)
    named = []
    for name, mod in net.named_modules():
        if isinstance(mod, (ConvModule, NonLocal3d)):
            named.append((name, mod.rec))
    pools = {}
    for name in ("maxpool", "pool2"):
        p = getattr(net, name)
        pools[name] = dict(kernel_size=p.kernel_size, stride=p.stride, padding=p.padding)
    out = dict(modules=named, pools=pools, feat_dim=net.feat_dim, final_feat_dim=net.final_feat_dim,
               res_layers=list(net.res_layers), num_stages=net.num_stages, with_pool2=net.with_pool2,
               backbone_kwargs={k: v for k, v in backbone.items() if k not in ("pretrained",)})
    np.savez_compressed(os.path.join(OUT, "py_i3d_wiring.npz"), wiring=jbytes(out))
    print("i3d wiring: %d ConvModule / NonLocal3d constructions, feat_dim %d" % (len(named), net.feat_dim))


def main():
    cfg = ref_env.setup()
    os.makedirs(OUT, exist_ok=True)
    gt_branch(cfg)
    i3d_wiring(cfg)
    for f in ("py_gt_branch.npz", "py_i3d_wiring.npz"):
        print("%-28s %8d bytes" % (f, os.path.getsize(os.path.join(OUT, f))))


if __name__ == "__main__":
    main()
