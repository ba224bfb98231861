"""Round-2 fixtures produced by the reference's own Python (tools/make_goldens_r2.py) against the oracle restatements
and the product's host logic (no GPU needed):
  py_gt_branch.npz   model_builder_rel.py:773-810, :845-858 (gt-box branch: pair / box de-duplication, gt labels)
  py_i3d_wiring.npz  constructor arguments of every ConvModule / NonLocal3d of lib/i3d/resnet3d.py under the DC5 overrides
"""
import json
import os

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _triple(v):
    return tuple(v) if isinstance(v, (list, tuple)) else (v, v, v)


def _cases(g):
    return json.loads(bytes(g["cases"]).decode())


@pytest.mark.parametrize("who", ["oracle", "product"])
def test_gt_box_pairs_match_reference_lines(who):
    from oracle import np_oracle
    g = np.load(os.path.join(GOLDEN, "py_gt_branch.npz"))
    for c in _cases(g):
        sb, ob = g[c + "_in_sbj_gt_boxes"], g[c + "_in_obj_gt_boxes"]
        sc, oc = g[c + "_in_sbj_gt_classes"], g[c + "_in_obj_gt_classes"]
        scale = g[c + "_in_im_info"][0, 2]
        if who == "oracle":
            out = np_oracle.gt_box_pairs(sb, ob, scale, sc, oc)
        else:
            from trace_amd.model_builder_rel import Generalized_RCNN
            out = Generalized_RCNN.gt_box_pairs(dict(sbj_gt_boxes=sb, obj_gt_boxes=ob, sbj_gt_classes=sc,
                                                     obj_gt_classes=oc), scale, use_gt_labels=True)
        for k in ("sbj_rois", "obj_rois", "rel_rois", "det_rois", "sbj_inds", "obj_inds", "pair_gt_index", "sbj_labels",
                  "obj_labels", "sbj_scores", "obj_scores", "det_labels", "det_scores"):
            want = g[c + "_out_" + k]
            got = np.asarray(out[k])
            assert got.shape == want.shape, (c, k, got.shape, want.shape)
            assert np.array_equal(got.astype(want.dtype), want), (c, k)
        # what the branch guarantees and the HIP path relies on: per-pair rois ARE rows of the de-duplicated list
        assert np.array_equal(out["sbj_rois"], out["det_rois"][out["sbj_inds"]])
        assert np.array_equal(out["obj_rois"], out["det_rois"][out["obj_inds"]])


def _wiring():
    g = np.load(os.path.join(GOLDEN, "py_i3d_wiring.npz"))
    return json.loads(bytes(g["wiring"]).decode())


def test_i3d_oracle_tables_match_reference_constructor_arguments():
    """oracle/i3d_ref.py's hard-coded tables (which blocks inflate / carry a non-local block, per-conv stride, padding
    and dilation) against what lib/i3d/resnet3d.py itself passes to ConvModule / NonLocal3d."""
    from oracle.i3d_ref import wiring
    w = _wiring()
    convs, nl, pools = wiring()
    ref_convs = {n: r for n, r in w["modules"] if r["kind"] == "ConvModule"}
    ref_nl = [n for n, r in w["modules"] if r["kind"] == "NonLocal3d"]
    assert sorted(ref_convs) == sorted(convs) and len(convs) == 53
    for name, (k, s, p, d) in convs.items():
        r = ref_convs[name]
        assert (_triple(r["kernel_size"]), _triple(r["stride"]), _triple(r["padding"]), _triple(r["dilation"])) == (k, s, p, d), name
        assert r["norm_type"] == "BN3d" and r["bias"] is False
        assert (r["act_type"] is None) == (name.endswith("conv3") or name.endswith("downsample")), name
    assert ref_nl == nl
    for name, r in w["modules"]:
        if r["kind"] == "NonLocal3d":
            assert r["kwargs"] == dict(sub_sample=True, use_scale=False, norm_cfg=dict(type="BN3d", requires_grad=True),
                                       mode="dot_product")
    for name, (k, s, p) in pools.items():
        assert (_triple(w["pools"][name]["kernel_size"]), _triple(w["pools"][name]["stride"]),
                _triple(w["pools"][name]["padding"])) == (k, s, p), name
    assert w["feat_dim"] == 2048 and w["num_stages"] == 4 and w["with_pool2"] is True


def test_product_i3d_matches_reference_constructor_arguments():
    """trace_amd.resnet3d.ResNet3d(**i3d_dc5_config()) builds the same convs (channels, kernel, stride, padding,
    dilation, bias / norm / activation) and non-local blocks, under the same module names."""
    from trace_amd.resnet3d import ConvModule, NonLocal3d, ResNet3d, i3d_dc5_config
    w = _wiring()
    m = ResNet3d(**i3d_dc5_config())
    mine = [(n, mod) for n, mod in m.named_modules()
            if (isinstance(mod, ConvModule) and "non_local_block" not in n) or isinstance(mod, NonLocal3d)]
    assert [n for n, _ in mine] == [n for n, _ in w["modules"]]
    for (name, mod), (_, r) in zip(mine, w["modules"]):
        if r["kind"] == "NonLocal3d":
            assert mod.in_channels == r["in_channels"] and mod.sub_sample and not mod.use_scale
            assert mod.conv_out.with_norm and not mod.theta.with_norm
            continue
        c = mod.conv
        assert (c.in_channels, c.out_channels) == (r["in_channels"], r["out_channels"]), name
        assert (tuple(c.kernel_size), tuple(c.stride), tuple(c.padding), tuple(c.dilation)) == (
            _triple(r["kernel_size"]), _triple(r["stride"]), _triple(r["padding"]), _triple(r["dilation"])), name
        assert (c.bias is not None) == bool(r["bias"]) and mod.with_norm == (r["norm_type"] is not None), name
        assert mod.with_activation == (r["act_type"] is not None), name
    assert tuple(m.maxpool.kernel_size) == _triple(w["pools"]["maxpool"]["kernel_size"])
    assert tuple(m.maxpool.stride) == _triple(w["pools"]["maxpool"]["stride"])
    assert tuple(m.pool2.kernel_size) == _triple(w["pools"]["pool2"]["kernel_size"])
    assert m.feat_dim == w["feat_dim"] and m.final_feat_dim == w["final_feat_dim"] and m.res_layers == w["res_layers"]
    kw = i3d_dc5_config()
    for k in ("depth", "num_stages", "inflate", "non_local", "dilations", "spatial_strides", "zero_init_residual", "norm_eval"):
        ref_v = w["backbone_kwargs"][k]
        mine_v = kw[k]
        assert json.loads(json.dumps(mine_v)) == ref_v, k
