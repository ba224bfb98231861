"""The CPU oracle (oracle/np_oracle.py, oracle/head_ref.py) against golden vectors produced by RUNNING
THE REFERENCE'S OWN PYTHON in the build container (tools/make_goldens_py.py, environment tools/ref_env.py).

Integer / index / box arithmetic: bit-exact.  fp32 network stages: 1e-5 relative (same torch ops, the
oracle groups a few sums differently, e.g. the tree-GRU's dense-mask products)."""
import json

import numpy as np
import pytest
import torch

from oracle import np_oracle, head_ref


def jload(a):
    return json.loads(bytes(a.tobytes()).decode())


def layers_from_json(obj):
    return [{int(fa): [int(s) for s in sons] for fa, sons in layer} for layer in obj]


def sd_of(g, prefix=""):
    return {k[2 + len(prefix):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w:" + prefix)}


def close(a, b, rtol=1e-5):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    denom = max(1e-6, float(np.abs(b).max()))
    err = float(np.abs(a - b).max()) / denom
    assert err <= rtol, err


# ----------------------------------------------------------------------------------------------- box glue
def test_fpn_levels_and_blobs_match_reference(golden):
    g = golden("py_box_glue.npz")
    boxes = g["fpn_boxes"]
    for lo, hi in ((2, 5), (3, 4)):
        lv = np_oracle.map_rois_to_fpn_levels(boxes.copy(), lo, hi)
        ref = g["fpn_levels_%d_%d" % (lo, hi)]
        assert lv.dtype == ref.dtype and np.array_equal(lv, ref)
    rois = np.concatenate([np.zeros((boxes.shape[0], 1), np.float32), boxes], 1)
    blobs = {}
    np_oracle.add_multilevel_roi_blobs(blobs, "rois", rois, g["fpn_levels_2_5"], 2, 5)
    for k, v in blobs.items():
        ref = g["blob_" + k]
        assert v.dtype == ref.dtype and np.array_equal(v, ref), k
    areas, neg = np_oracle.boxes_area(boxes.copy())
    assert np.array_equal(areas, g["areas"]) and np.array_equal(neg, g["areas_neg_idx"])


def test_bbox_transform_clip_match_reference(golden):
    g = golden("py_box_glue.npz")
    pred = np_oracle.bbox_transform(g["bt_boxes"], g["bt_deltas"], tuple(g["bt_weights"].tolist()),
                                    float(g["bt_xform_clip"]))
    assert pred.dtype == g["bt_pred"].dtype and np.array_equal(pred, g["bt_pred"])
    clipped = np_oracle.clip_tiled_boxes(pred.copy(), tuple(g["bt_im_shape"].tolist()))
    assert np.array_equal(clipped, g["bt_clipped"])


def test_rois_union_and_argsort_desc_match_reference(golden):
    g = golden("py_box_glue.npz")
    assert np.array_equal(np_oracle.rois_union(g["union_a"], g["union_b"]), g["union_out"])
    assert np.array_equal(np_oracle.argsort_desc(g["asd_scores"]), g["asd_out"])
    # the ranking core built on it (task_evaluation_vg_and_vrd.py:207-228) picks the same (pair, predicate) cells
    sc = g["asd_scores"]
    s, pair, prd = np_oracle.topk_triplets(sc, np.ones(sc.shape[0], np.float32), np.ones(sc.shape[0], np.float32),
                                           sc.shape[1], 10)
    top = g["asd_out"][:10]
    assert np.allclose(s, sc[top[:, 0], top[:, 1]])


# ----------------------------------------------------------------------------------------------- tree
def test_node_cluster_adjacency_lca_match_reference(golden):
    g = golden("py_tree.npz")
    for key in jload(g["cases"]):
        boxes = g[key + "_boxes"]
        n = boxes.shape[0]
        layers, tree = np_oracle.node_cluster(boxes, 7, 2)
        ref_layers = layers_from_json(jload(g[key + "_layers"]))
        assert [{k: list(map(int, v)) for k, v in l.items()} for l in layers] == ref_layers, key
        ref_tree = g[key + "_tree_bbox"]                     # reference carries the batch id in column 0
        assert np.array_equal(tree.astype(ref_tree.dtype), ref_tree[:, -4:]), key
        A, A_T, link = np_oracle.get_A_and_link(tree.shape[0], layers)
        assert np.array_equal(A, g[key + "_A"]) and np.array_equal(A_T, g[key + "_A_T"]), key
        assert np.array_equal(link, g[key + "_link"]), key
        tnl = np_oracle.temporal_nonleaf_rois(layers, tree.shape[0], n, g[key + "_tboxes"])
        assert np.array_equal(tnl, g[key + "_tnonleaf"]), key
        if n > 1:
            lca = np_oracle.get_lca(g[key + "_X"], g[key + "_Y"], link)
            assert np.array_equal(lca, g[key + "_lca"]), key


def test_relpn_pairs_match_reference(golden):
    g = golden("py_relpn.npz")
    for key in jload(g["cases"]):
        det_rois, labels, scores = g[key + "_det_rois"], g[key + "_labels"], g[key + "_scores"]
        s, o, rel = np_oracle.relpn_pairs(det_rois, labels, jload(g[key + "_dataset"]))
        assert np.array_equal(s, g[key + "_out_sbj_inds"]) and np.array_equal(o, g[key + "_out_obj_inds"]), key
        assert np.array_equal(rel, g[key + "_out_rel_rois"]), key
        assert np.array_equal(det_rois[s], g[key + "_out_sbj_rois"])
        assert np.array_equal(labels[s], g[key + "_out_sbj_labels"]) and np.array_equal(labels[o], g[key + "_out_obj_labels"])
        assert np.array_equal(scores[s], g[key + "_out_sbj_scores"])
        assert np.array_equal(labels[o].astype(np.int32) - 1, g[key + "_out_all_obj_labels_int32"])


# ----------------------------------------------------------------------------------------------- network stages
def test_box_heads_match_reference(golden):
    g = golden("py_heads.npz")
    sd = sd_of(g)
    pooled = head_ref._roi_align(torch.from_numpy(g["blob_conv"]), g["det_rois"], 1.0 / 16, 2)
    close(head_ref._two_mlp(sd, "Box_Head", pooled, True), g["vec_relu"])
    close(head_ref._two_mlp(sd, "Box_Head", pooled, False), g["vec_lin"])
    v = torch.from_numpy(g["vec_relu"])
    close(torch.softmax(head_ref._lin(sd, "Box_Outs.cls_score", v), 1), g["cls_score"])
    close(head_ref._lin(sd, "Box_Outs.bbox_pred", v), g["bbox_pred"])
    t_in = torch.from_numpy(g["tail_in"])
    close(head_ref._tail(sd, "tail", t_in.reshape(-1, *t_in.shape[2:])), g["tail_out"])


def test_cluster_tree3d_forward_matches_reference(golden):
    g = golden("py_cluster_tree3d.npz")
    C, feat_vec, combine, low_dim, groups, Tn = g["dims"].tolist()
    sd = {"cluster_tree3d." + k: v for k, v in sd_of(g, "cluster_tree3d.").items()}
    out = head_ref.cluster_tree3d_forward(sd, "cluster_tree3d", g["roi"], torch.from_numpy(g["feat_map"]),
                                          torch.from_numpy(g["obj_feat_vec"]), torch.from_numpy(g["tmaps"][0]),
                                          low_dim=low_dim, groups=groups)
    ref_layers = layers_from_json(jload(g["out_layers"]))
    assert [{k: list(map(int, v)) for k, v in l.items()} for l in out["layers"]] == ref_layers
    assert np.array_equal(out["A"], g["out_A"]) and np.array_equal(out["link"], g["out_link"])
    assert np.array_equal(out["nonleaf_rois"], g["out_nonleaf_bbox"])
    assert np.array_equal(out["region_map"].numpy(), g["out_region_map"])          # pinned C ROIAlign, bit-exact
    close(out["pooled_features"], g["out_pooled_features"])
    s, o = torch.from_numpy(g["tatt_s"]), torch.from_numpy(g["tatt_o"])
    close(head_ref._temporal_attention(sd, "cluster_tree3d.temporal_obj_atten", s, o, low_dim), g["tatt_obj_out"])
    close(head_ref._temporal_attention(sd, "cluster_tree3d.temporal_r_atten", s, o, low_dim), g["tatt_r_out"])


def test_pair_attention_and_sem_match_reference(golden):
    g = golden("py_pair_head.npz")
    sd = sd_of(g)
    roi = g["roi"]
    s_map, o_map, v_map, lca, union, _, _ = head_ref.pair_attention(
        sd, roi, torch.from_numpy(g["feat_map"]), torch.from_numpy(g["obj_feature"]), g["link"], g["X"], g["Y"])
    assert np.array_equal(lca, g["out_lca"])
    close(v_map, g["out_v_map"])
    close(s_map, g["out_s_map"])
    close(o_map, g["out_o_map"])
    sem = head_ref.sem_feat(sd, torch.from_numpy(g["softmax_obj_vec"]), torch.from_numpy(g["region_feature"]), lca,
                            g["X"], g["Y"])
    close(sem, g["out_sem"])


def test_det_branch_matches_reference(golden):
    """prepare_det_rois / get_det_boxes / box_results_with_nms_and_limit (model_builder_rel.py:1212-1340)."""
    g = golden("py_det_branch.npz")
    c = jload(g["cfg"])
    for key in jload(g["cases"]):
        dr, dl, ds, di = np_oracle.prepare_det_rois(g[key + "_rois"], g[key + "_cls_scores"], g[key + "_bbox_pred"],
                                                    g[key + "_im_info"], 1, c["score_thresh"], c["nms"],
                                                    c["detections_per_im"], tuple(c["reg_weights"]))
        assert np.array_equal(di, g[key + "_out_inds"]), key
        assert np.array_equal(dl, g[key + "_out_det_labels"]), key
        assert np.array_equal(ds, g[key + "_out_det_scores"]), key
        assert np.array_equal(dr, g[key + "_out_det_rois"]), key
    assert g["r300_out_inds"].shape[0] == 100            # the DETECTIONS_PER_IM cap was exercised
