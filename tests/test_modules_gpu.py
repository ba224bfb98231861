"""Module-level boundary: the head modules called with the reference's own constructor arguments and
`forward` signatures, checked DIRECTLY against outputs of the reference's modules (tests/golden/py_heads.npz,
py_cluster_tree3d.npz, produced by tools/make_goldens_py.py from lib/modeling_rel/fast_rcnn_heads.py,
lib/i3d/box_head3d.py, lib/combine_tree/cluster_tree3d.py at reduced widths).  bf16 GEMMs vs the reference's fp32:
features within 2e-2 (north_star); everything integer / structural and the f32 ROIAlign maps bit-exact."""
import json
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel_l2(got, ref):
    got, ref = torch.as_tensor(got).float().cpu(), torch.as_tensor(ref).float().cpu()
    return float((got - ref).norm() / (ref.norm() + 1e-12))


def small_cfg():
    from trace_amd.config import TraceConfig
    return TraceConfig(NUM_CLASSES=9, MLP_HEAD_DIM=64, RESIZE_HIDDEN_DIM=2, LOW_FEAT_DIM=16, MSP_GROUP=4)


def load(module, g, prefix):
    sd = {k[len("w:" + prefix):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w:" + prefix)}
    missing = module.load_state_dict(sd, strict=True)
    return module.to(DEV).eval()


def test_roi_2mlp_head_outputs_and_tail_match_reference_modules():
    from trace_amd.heads import Tail_Layer_3D, fast_rcnn_outputs, roi_2mlp_head
    from trace_amd.roi_xform import roi_feature_transform
    g = np.load(os.path.join(GOLDEN, "py_heads.npz"))
    cfg = small_cfg()
    C = g["blob_conv"].shape[1]
    head = load(roi_2mlp_head(C, roi_feature_transform, 1.0 / 16, cfg=cfg), g, "Box_Head.")      # reference argument order
    outs = load(fast_rcnn_outputs(head.dim_out, cfg), g, "Box_Outs.")
    host = types.SimpleNamespace(final_feat_dim=C)
    tail = load(Tail_Layer_3D(host, [], None, 1.0 / 16, num_stages=4, cfg=cfg), g, "tail.")
    blob = torch.from_numpy(g["blob_conv"]).to(DEV)
    rpn_ret = {"rois": g["det_rois"]}
    v_relu = head(blob, rpn_ret, rois_name="rois", use_relu=True)
    v_lin = head(blob, rpn_ret, rois_name="rois", use_relu=False)
    assert v_relu.dtype == torch.float32 and tuple(v_relu.shape) == g["vec_relu"].shape
    assert rel_l2(v_relu, g["vec_relu"]) < 2e-2 and rel_l2(v_lin, g["vec_lin"]) < 2e-2
    # the pooled map itself is the f32 kernel: bit-identical to the reference's CPU ROIAlign
    pooled = head.roi2map(blob, rpn_ret)
    from oracle import c_oracle
    assert np.array_equal(pooled.cpu().numpy(), c_oracle.roi_align_fwd(g["blob_conv"], g["det_rois"], 1 / 16., 7, 7, 2))
    cls, bbox = outs(torch.from_numpy(g["vec_relu"]).to(DEV))
    assert rel_l2(cls, g["cls_score"]) < 2e-2 and rel_l2(bbox, g["bbox_pred"]) < 2e-2
    assert np.allclose(cls.sum(1).cpu().numpy(), 1.0, atol=1e-4)                 # eval mode: softmax (:46-48)
    t_out = tail(torch.from_numpy(g["tail_in"]).to(DEV))
    assert tuple(t_out.shape) == g["tail_out"].shape and rel_l2(t_out, g["tail_out"]) < 2e-2
    # channels-last bf16 entry of the same head (what the fused pipeline feeds it)
    v2 = head(blob.permute(0, 2, 3, 1).contiguous().bfloat16(), rpn_ret, use_relu=True)
    assert rel_l2(v2, g["vec_relu"]) < 2e-2


def test_cluster_tree3d_forward_13_tuple_matches_reference_module():
    from trace_amd.heads import Tail_Layer_3D, cluster_tree3d
    from trace_amd.roi_layers import ROIAlign
    g = np.load(os.path.join(GOLDEN, "py_cluster_tree3d.npz"))
    C, feat_vec, combine, low_dim, groups, Tn = (int(v) for v in g["dims"])
    cfg = small_cfg()
    host = types.SimpleNamespace(final_feat_dim=C)
    ct3 = cluster_tree3d((C, 0, 0), combine, feat_vec, feat_vec, combine, low_dim, ROIAlign, 1.0 / 16,
                         group_coeffi=groups, max_tree_height=7, pool_size=7,
                         BoxHead=Tail_Layer_3D(host, [], None, 1.0 / 16, num_stages=4, cfg=cfg), cluster_num_partition=2,
                         msp_mode="tree-gru")
    ct3 = load(ct3, g, "cluster_tree3d.")
    roi, troi = g["roi"], g["temporal_roi"]
    fm = torch.from_numpy(g["feat_map"]).to(DEV)
    tm = torch.from_numpy(g["tmaps"]).to(DEV)
    res = ct3(roi, fm, torch.from_numpy(g["obj_feat_vec"]).to(DEV), troi, tm, tm)
    assert len(res) == 13
    (pooled_features, region_map, layer_list, tree_nl, tree_ns, leaf_nl, leaf_ns, nonleaf_n, nonleaf_ns, bA, bAT, blink,
     nonleaf_bbox) = res
    n = roi.shape[0]
    n_tree = g["out_A"].shape[0]
    assert tree_nl == [n_tree] and tree_ns == [0, n_tree] and leaf_nl == [n] and leaf_ns == [0, n]
    assert nonleaf_n == [n_tree - n] and nonleaf_ns == [0, n_tree - n]
    want_layers = json.loads(bytes(g["out_layers"]).decode())
    got_layers = [[[int(fa), [int(s) for s in sons]] for fa, sons in layer.items()] for layer in layer_list[0]]
    assert got_layers == want_layers
    assert np.array_equal(bA[0], g["out_A"]) and np.array_equal(bAT[0], g["out_A_T"])
    assert np.array_equal(np.asarray(blink[0]).astype(np.int64), g["out_link"])
    assert np.array_equal(np.asarray(nonleaf_bbox, np.float32), g["out_nonleaf_bbox"])
    assert np.array_equal(region_map.cpu().numpy(), g["out_region_map"])          # f32 ROIAlign: bit-exact
    assert tuple(pooled_features.shape) == g["out_pooled_features"].shape
    e = rel_l2(pooled_features, g["out_pooled_features"])
    assert e < 2e-2, e
    # the two temporal attentions as methods (cluster_tree3d.py:337-366)
    s, o = torch.from_numpy(g["tatt_s"]).to(DEV), torch.from_numpy(g["tatt_o"]).to(DEV)
    assert rel_l2(ct3.temporal_obj_attention(s, o), g["tatt_obj_out"]) < 2e-2
    assert rel_l2(ct3.temporal_r_attention(s, o), g["tatt_r_out"]) < 2e-2


def test_roi_feature_transform_fpn_dispatch_and_restore():
    """The FPN path of roi_feature_transform (model_builder_rel.py:1354-1382, cluster_tree.py:700-726): per-level
    ROIAlign, concatenation, un-shuffle by idx_restore - on the level blobs the REFERENCE's add_multilevel_roi_blobs
    produced (py_box_glue.npz).  Row i must equal ROIAlign of roi i on its own level's map, bit for bit."""
    from oracle import c_oracle
    from trace_amd.roi_xform import add_rel_multilevel_rois, roi_feature_transform
    g = np.load(os.path.join(GOLDEN, "py_box_glue.npz"))
    boxes = g["fpn_boxes"]
    rois = np.concatenate([np.zeros((boxes.shape[0], 1), np.float32), boxes], 1)
    rng = np.random.default_rng(2)
    k_min, k_max = 2, 5
    maps = {lvl: rng.standard_normal((1, 8, 1400 // 2 ** lvl + 1, 1400 // 2 ** lvl + 1)).astype(np.float32)
            for lvl in range(k_min, k_max + 1)}
    blobs_in = [torch.from_numpy(maps[lvl]).to(DEV) for lvl in range(k_max, k_min - 1, -1)]      # reversed order (:1362)
    scales = [1.0 / 2 ** lvl for lvl in range(k_max, k_min - 1, -1)]
    rpn_ret = {"rois": rois}
    for lvl in range(k_min, k_max + 1):
        rpn_ret["rois_fpn%d" % lvl] = g["blob_rois_fpn%d" % lvl]
    rpn_ret["rois_idx_restore_int32"] = g["blob_rois_idx_restore_int32"]
    out = roi_feature_transform(blobs_in, rpn_ret, blob_rois="rois", method="RoIAlign", resolution=7,
                                spatial_scale=scales, sampling_ratio=2).cpu().numpy()
    lv = g["fpn_levels_2_5"].astype(int)
    assert out.shape == (rois.shape[0], 8, 7, 7)
    for i in range(rois.shape[0]):
        want = c_oracle.roi_align_fwd(maps[lv[i]], rois[i:i + 1], 1.0 / 2 ** lv[i], 7, 7, 2)
        assert np.array_equal(out[i:i + 1], want), i
    # the product's own level split (host float32 arithmetic) gives the reference's blobs
    mine = add_rel_multilevel_rois({"rois": rois}, "rois", k_min, k_max)
    for k in rpn_ret:
        assert np.array_equal(mine[k], rpn_ret[k]), k
    # ROIPool through the same dispatch
    outp = roi_feature_transform(blobs_in, rpn_ret, blob_rois="rois", method="RoIPoolF", resolution=7,
                                 spatial_scale=scales)
    assert tuple(outp.shape) == (rois.shape[0], 8, 7, 7)


def test_reloading_weights_through_the_parent_refolds_the_stem():
    """forward, then model.load_state_dict(other checkpoint) through the PARENT module, then forward: every packed
    weight (stem included) and every captured graph must follow - compared with a freshly built model."""
    from trace_amd.model_builder_rel import Generalized_RCNN
    from trace_amd.synthetic import make_inputs, randomize_model

    def build(seed):
        torch.manual_seed(seed)
        m = Generalized_RCNN()
        randomize_model(m, seed)
        return m.eval()
    frames, blob, feat, rois = make_inputs(1, 8, 5, 224, seed=9)
    args = (frames.to(DEV), blob.to(DEV), feat.to(DEV), rois)
    a, b = build(1).to(DEV), build(2)
    a.graph_step = True
    first = a.forward_clips(*args)[0]["prd_scores"].clone()
    a.load_state_dict(b.state_dict())                    # nn.Module.load_state_dict recursion: children's own overrides never run
    got = a.forward_clips(*args)[0]["prd_scores"].clone()
    want = b.to(DEV).forward_clips(*args)[0]["prd_scores"]
    assert not torch.equal(first, got)
    assert torch.equal(got, want)
    # a checkpoint loaded into ONE sub-module
    c = build(3)
    a.Box_Head.load_state_dict(c.Box_Head.state_dict())
    a.Prd_RCNN.conv1.load_state_dict(c.Prd_RCNN.conv1.state_dict())
    b.Box_Head.load_state_dict(c.Box_Head.state_dict())
    b.Prd_RCNN.conv1.load_state_dict(c.Prd_RCNN.conv1.state_dict())
    assert torch.equal(a.forward_clips(*args)[0]["prd_scores"], b.forward_clips(*args)[0]["prd_scores"])
