"""Generate tests/golden/py_*.npz by RUNNING THE REFERENCE'S OWN PYTHON (imported from
/root/reference on the CPU of the build container; environment in tools/ref_env.py).  Fixtures hold
inputs, seeded weights and the reference's outputs - data only, no reference text.

    python tools/make_goldens_py.py

What is pinned (each checked against the oracle restatement in tests/test_oracle_goldens_py.py):
  py_box_glue.npz        utils/fpn.py:11-58, utils/boxes.py:58-73,142-200, utils_rel/boxes_rel.py:100-106,
                         datasets_rel/pytorch_misc.py:28-35
  py_tree.npz            combine_tree/cluster_tree.py:504-649 node_cluster, :400-430 get_A_and_link,
                         modeling_rel/model_builder_rel.py:1614-1619 get_lca,
                         combine_tree/cluster_tree3d.py:368-391 temporal_get_new_bbox_feat
  py_relpn.npz           modeling_rel/relpn_heads.py:57-130 (eval mode, no overlap filter: it needs Cython)
  py_heads.npz           modeling_rel/fast_rcnn_heads.py roi_2mlp_head / fast_rcnn_outputs, i3d/box_head3d.py Tail_Layer_3D
  py_cluster_tree3d.npz  combine_tree/cluster_tree3d.py:114-366 whole forward (temporal ROIAlign, attention,
                         region tree, tree-GRU x groups, FFN) at reduced widths
  py_pair_head.npz       model_builder_rel.py:1524-1612 get_spatio_atten_feat, :1643-1668 get_sem_feat
  py_det_branch.npz      model_builder_rel.py:1212-1340 prepare_det_rois / get_det_boxes / box_results_with_nms_and_limit
"""
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_env  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def synth_boxes(rng, n, k=2.0, size=224.0):
    x1 = rng.uniform(0, 160, n)
    y1 = rng.uniform(0, 160, n)
    w = rng.uniform(16, 64 * k, n)
    h = rng.uniform(16, 64 * k, n)
    return np.stack([x1, y1, np.minimum(x1 + w, size - 1), np.minimum(y1 + h, size - 1)], 1).astype(np.float32)


def jbytes(obj):
    return np.frombuffer(json.dumps(obj).encode(), dtype=np.uint8)


def layers_to_json(layer_list):
    return [[[int(fa), [int(s) for s in sons]] for fa, sons in layer.items()] for layer in layer_list]


def sd_np(module, prefix=""):
    return {"w:" + prefix + k: v.detach().numpy().copy() for k, v in module.state_dict().items()}


def randomize(module, gen, scale=1.0):
    """Seeded weights (the reference's init schemes leave biases at 0, which would hide bias handling)."""
    with torch.no_grad():
        for name, p in module.named_parameters():
            if p.dim() >= 2:
                fan_in = p[0].numel()
                p.copy_(torch.randn(p.shape, generator=gen) * (scale / np.sqrt(fan_in)))
            elif name.endswith("weight"):          # LayerNorm gain
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=gen))
            else:
                p.copy_(0.1 * torch.randn(p.shape, generator=gen))


def main():
    cfg = ref_env.setup()
    import utils.fpn as fpn_utils
    import utils.boxes as box_utils
    import utils_rel.boxes_rel as box_utils_rel
    from datasets_rel.pytorch_misc import argsort_desc
    from combine_tree.cluster_tree3d import cluster_tree3d
    from model.roi_layers import ROIAlign
    from modeling_rel import relpn_heads, fast_rcnn_heads
    from modeling_rel.model_builder_rel import Generalized_RCNN
    from i3d.box_head3d import Tail_Layer_3D
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(11)
    gen = torch.Generator().manual_seed(11)

    # ------------------------------------------------------------------ box glue
    g = {}
    boxes = np.concatenate([synth_boxes(rng, 40, 6.0, 1400.0),
                            np.array([[0, 0, 223, 223], [0, 0, 111, 111], [0, 0, 447, 447], [0, 0, 2000, 2000],
                                      [0, 0, 5, 5], [10, 10, 5, 5], [30, 30, 20, 80], [0, 0, 0, 0]], np.float32)], 0)
    g["fpn_boxes"] = boxes
    g["fpn_levels_2_5"] = fpn_utils.map_rois_to_fpn_levels(boxes.copy(), 2, 5)
    g["fpn_levels_3_4"] = fpn_utils.map_rois_to_fpn_levels(boxes.copy(), 3, 4)
    rois = np.concatenate([np.zeros((boxes.shape[0], 1), np.float32), boxes], 1)
    blobs = {}
    fpn_utils.add_multilevel_roi_blobs(blobs, "rois", rois, g["fpn_levels_2_5"], 2, 5)
    for k, v in blobs.items():
        g["blob_" + k] = v
    areas, neg = box_utils.boxes_area(boxes.copy())
    g["areas"], g["areas_neg_idx"] = areas, neg
    R, K = 30, 5
    b0 = synth_boxes(rng, R, 3.0, 600.0)
    deltas = (rng.standard_normal((R, 4 * K)) * np.array([1.5, 1.5, 4.0, 4.0] * K)).astype(np.float32)
    deltas[0, 2] = 80.0      # dw beyond BBOX_XFORM_CLIP
    g["bt_boxes"], g["bt_deltas"] = b0, deltas
    g["bt_weights"] = np.array(cfg.MODEL.BBOX_REG_WEIGHTS, np.float32)
    g["bt_xform_clip"] = np.float64(cfg.BBOX_XFORM_CLIP)
    pred = box_utils.bbox_transform(b0, deltas, cfg.MODEL.BBOX_REG_WEIGHTS)
    g["bt_pred"] = pred
    g["bt_im_shape"] = np.array([480, 600], np.int64)
    g["bt_clipped"] = box_utils.clip_tiled_boxes(pred.copy(), (480, 600))
    r1 = np.concatenate([np.zeros((12, 1), np.float32), synth_boxes(rng, 12)], 1)
    r2 = np.concatenate([np.zeros((12, 1), np.float32), synth_boxes(rng, 12)], 1)
    g["union_a"], g["union_b"], g["union_out"] = r1, r2, box_utils_rel.rois_union(r1, r2)
    sc = rng.random((17, 5)).astype(np.float32)
    g["asd_scores"], g["asd_out"] = sc, argsort_desc(sc)
    np.savez_compressed(os.path.join(OUT, "py_box_glue.npz"), **g)

    # ------------------------------------------------------------------ widths of the reduced model
    C, H, W, Tn = 32, 14, 14, 3
    cfg.FAST_RCNN.MLP_HEAD_DIM = 64
    cfg.FAST_RCNN.ROI_XFORM_RESOLUTION = 7
    cfg.FAST_RCNN.ROI_XFORM_SAMPLING_RATIO = 2
    cfg.FAST_RCNN.ROI_XFORM_METHOD = "RoIAlign"
    cfg.I3D_DC5 = True
    cfg.NO_TEMPORAL_FUSION = False
    cfg.MODEL.NUM_CLASSES = 9
    feat_vec, combine, low_dim, groups = 64, 32, 16, 4
    scale = 1.0 / 16
    host = types.SimpleNamespace(final_feat_dim=C)          # the two attributes Tail_Layer_3D reads of the I3D net
    ct = cluster_tree3d((C, 0, 0), combine, feat_vec, feat_vec, combine, low_dim, ROIAlign, scale,
                        group_coeffi=groups, max_tree_height=7, pool_size=7,
                        BoxHead=Tail_Layer_3D(host, [], None, scale, num_stages=4), cluster_num_partition=2,
                        msp_mode="tree-gru")
    randomize(ct, gen, 1.5)
    ct.eval()

    # ------------------------------------------------------------------ tree building
    t = {}
    cases = []
    for n in (1, 2, 3, 5, 10, 21, 50):
        bx = synth_boxes(rng, n)
        roi = np.concatenate([np.zeros((n, 1), np.float32), bx], 1)
        _, nl, ns = ct.get_bbox_id(roi, 1)
        layer_list, tree_bbox = ct.node_cluster(roi[:, 1:5], 1, nl, ns, dist_type="euclid", feat_type="coordinate",
                                                add_batch_bbox_feat=None)
        _, tl, ts = ct.get_bbox_id(tree_bbox, 1)
        A, A_T, link = ct.get_A_and_link(1, tl, layer_list, torch.device("cpu"))
        tb = np.repeat(bx[:, None, :], Tn, 1) + rng.uniform(-3, 3, (n, Tn, 4)).astype(np.float32)
        tnl = ct.temporal_get_new_bbox_feat(layer_list, ts, nl, ns, tb, 1, feat_type="coordinate")
        key = "n%d" % n
        cases.append(key)
        t[key + "_boxes"] = bx
        t[key + "_layers"] = jbytes(layers_to_json(layer_list[0]))
        t[key + "_tree_bbox"] = np.asarray(tree_bbox)
        t[key + "_A"], t[key + "_A_T"], t[key + "_link"] = A[0], A_T[0], link[0].astype(np.int64)
        t[key + "_tboxes"], t[key + "_tnonleaf"] = tb, tnl
        if n > 1:
            X = np.repeat(np.arange(n), n)
            Y = np.tile(np.arange(n), n)
            keep = X != Y
            X, Y = X[keep], Y[keep]
            t[key + "_X"], t[key + "_Y"] = X, Y
            t[key + "_lca"] = Generalized_RCNN.get_lca(None, X, Y, link[0]).astype(np.int64)
    t["cases"] = jbytes(cases)
    np.savez_compressed(os.path.join(OUT, "py_tree.npz"), **t)

    # ------------------------------------------------------------------ RelPN pair enumeration (eval)
    cfg.MODEL.USE_OVLP_FILTER = False
    cfg.MODEL.USE_SPATIAL_FEAT = False
    cfg.MODEL.USE_FREQ_BIAS = True
    cfg.FPN.FPN_ON = False
    cfg.TEST.DATASETS = ("vidvrd_val",)
    relpn = relpn_heads.single_scale_relpn_outputs()
    relpn.eval()
    r = {}
    rcases = []
    for name, n, ds in (("vidvrd_n1", 1, "vidvrd_val"), ("vidvrd_n2", 2, "vidvrd_val"), ("vidvrd_n7", 7, "vidvrd_val"),
                        ("ag_n6", 6, "ag_val"), ("ag_nopers_n4", 4, "ag_val")):
        bx = synth_boxes(rng, n) * 2.0
        det_rois = np.concatenate([np.zeros((n, 1), np.float32), bx], 1)
        labels = rng.integers(2, 9, n).astype(np.int32)
        if name == "ag_n6":
            labels[[1, 4]] = 1
        scores = rng.random(n).astype(np.float32)
        im_info = torch.tensor([[448.0, 448.0, 2.0]])
        out = relpn(det_rois, labels, scores, im_info, ds, roidb=None)
        rcases.append(name)
        r[name + "_det_rois"], r[name + "_labels"], r[name + "_scores"] = det_rois, labels, scores
        r[name + "_dataset"] = jbytes(ds)
        for k in ("sbj_inds", "obj_inds", "sbj_rois", "obj_rois", "rel_rois", "sbj_labels", "obj_labels",
                  "sbj_scores", "obj_scores", "all_sbj_labels_int32", "all_obj_labels_int32"):
            r[name + "_out_" + k] = np.asarray(out[k])
    r["cases"] = jbytes(rcases)
    np.savez_compressed(os.path.join(OUT, "py_relpn.npz"), **r)

    # ------------------------------------------------------------------ box heads
    def xform(im, rpn_ret, blob_rois="rois", method="RoIAlign", resolution=7, spatial_scale=1. / 16, sampling_ratio=0):
        return Generalized_RCNN.roi_feature_transform(None, im, rpn_ret, blob_rois, method, resolution, spatial_scale,
                                                      sampling_ratio)
    h = {}
    n = 6
    blob_conv = torch.randn((1, C, H, W), generator=gen)
    det = np.concatenate([np.zeros((n, 1), np.float32), synth_boxes(rng, n)], 1)
    head = fast_rcnn_heads.roi_2mlp_head(C, xform, scale)
    outs = fast_rcnn_heads.fast_rcnn_outputs(head.dim_out)
    randomize(head, gen)
    randomize(outs, gen, 3.0)
    head.eval()
    outs.eval()
    with torch.no_grad():
        v_relu = head(blob_conv, {"rois": det}, rois_name="rois", use_relu=True)
        v_lin = head(blob_conv, {"rois": det}, rois_name="rois", use_relu=False)
        cls, bbox = outs(v_relu)
        tail = Tail_Layer_3D(host, [], None, scale, num_stages=4)
        randomize(tail, gen)
        pooled = torch.randn((n, Tn, C, 7, 7), generator=gen)
        tail_out = tail(pooled)
    h.update(sd_np(head, "Box_Head."))
    h.update(sd_np(outs, "Box_Outs."))
    h.update(sd_np(tail, "tail."))
    h.update(blob_conv=blob_conv.numpy(), det_rois=det, vec_relu=v_relu.numpy(), vec_lin=v_lin.numpy(),
             cls_score=cls.numpy(), bbox_pred=bbox.numpy(), tail_in=pooled.numpy(), tail_out=tail_out.numpy())
    np.savez_compressed(os.path.join(OUT, "py_heads.npz"), **h)

    # ------------------------------------------------------------------ cluster_tree3d.forward, one clip
    c = {}
    n = 9
    bx = synth_boxes(rng, n)
    roi = np.concatenate([np.zeros((n, 1), np.float32), bx], 1)
    feat_map = torch.randn((1, C, H, W), generator=gen)
    tmaps = torch.randn((1, Tn, C, H, W), generator=gen)
    obj_vec = torch.randn((n, feat_vec), generator=gen)
    troi = np.repeat(roi[:, None, :], Tn, 1)                       # static tubes, model_builder_rel.py:1441-1444
    with torch.no_grad():
        res = ct(roi, feat_map, obj_vec, troi, tmaps, tmaps)
    (pooled_features, region_map, layer_list, tree_nl, tree_ns, leaf_nl, leaf_ns, nonleaf_n, nonleaf_ns,
     bA, bAT, blink, nonleaf_bbox) = res
    c.update(sd_np(ct, "cluster_tree3d."))
    c.update(roi=roi, feat_map=feat_map.numpy(), tmaps=tmaps.numpy(), obj_feat_vec=obj_vec.numpy(), temporal_roi=troi,
             out_pooled_features=pooled_features.numpy(), out_region_map=region_map.numpy(),
             out_layers=jbytes(layers_to_json(layer_list[0])), out_A=bA[0], out_A_T=bAT[0],
             out_link=blink[0].astype(np.int64), out_nonleaf_bbox=np.asarray(nonleaf_bbox),
             dims=np.array([C, feat_vec, combine, low_dim, groups, Tn], np.int64))
    # the temporal attention alone, larger T
    with torch.no_grad():
        s = torch.randn((5, feat_vec), generator=gen)
        o = torch.randn((5, 8, feat_vec), generator=gen)
        c["tatt_s"], c["tatt_o"] = s.numpy(), o.numpy()
        c["tatt_obj_out"] = ct.temporal_obj_attention(s, o).numpy()
        c["tatt_r_out"] = ct.temporal_r_attention(s, o).numpy()
    np.savez_compressed(os.path.join(OUT, "py_cluster_tree3d.npz"), **c)

    # ------------------------------------------------------------------ pair attention + semantic branch
    p = {}
    n_cls, embed, dim_out, infer_out = 9, 12, feat_vec, combine
    ns = types.SimpleNamespace()
    # layers built as model_builder_rel.py:259-263, 281-289, 302-312 (widths reduced)
    ns.spatio_atten_s = nn.Sequential(nn.LeakyReLU(0.1), nn.Linear(dim_out, low_dim, bias=True))
    ns.spatio_atten_o = nn.Sequential(nn.LeakyReLU(0.1), nn.Linear(dim_out, low_dim, bias=True))
    ns.spatio_atten_map = nn.Conv2d(C, low_dim, 1, stride=1, padding=0, bias=True)
    ns.atten_q_map = nn.Conv2d(C, low_dim, 1, stride=1, padding=0, bias=True)
    ns.embed_s = nn.Linear(n_cls, embed, bias=False)
    ns.embed_o = nn.Linear(n_cls, embed, bias=False)
    ns.sem_s_e = nn.Sequential(nn.LeakyReLU(0.1), nn.Linear(embed, infer_out, bias=True))
    ns.sem_o_e = nn.Sequential(nn.LeakyReLU(0.1), nn.Linear(embed, infer_out, bias=True))
    ns.sem_r = nn.Sequential(nn.LeakyReLU(0.1), nn.Linear(combine, infer_out, bias=True))
    for name in ("spatio_atten_s", "spatio_atten_o", "spatio_atten_map", "atten_q_map", "embed_s", "embed_o",
                 "sem_s_e", "sem_o_e", "sem_r"):
        m = getattr(ns, name)
        randomize(m, gen, 2.0)
        p.update(sd_np(m, name + "."))
    ns.pool_size = 7
    ns.rel_spatial_scale = scale
    for meth in ("get_lca", "get_rel_rois", "roi_feature_transform", "get_attention_mask_map"):
        setattr(ns, meth, types.MethodType(getattr(Generalized_RCNN, meth), ns))
    X = np.repeat(np.arange(n), n)
    Y = np.tile(np.arange(n), n)
    keep = X != Y
    X, Y = X[keep], Y[keep]
    obj_feature = torch.randn((n, dim_out), generator=gen)
    m_reg = tree_nl[0] - n
    region_feature = torch.randn((m_reg, combine), generator=gen)
    soft = torch.softmax(torch.randn((n, n_cls), generator=gen), 1)
    with torch.no_grad():
        s_map, o_map, v_map, lca_list = Generalized_RCNN.get_spatio_atten_feat(
            ns, roi, feat_map, obj_feature, region_map, leaf_nl, leaf_ns, blink, 1, nonleaf_ns, X, Y)
        sem = Generalized_RCNN.get_sem_feat(ns, obj_feature, region_feature, lca_list, X, Y, soft, 1, "vidvrd_val")
    p.update(roi=roi, feat_map=feat_map.numpy(), obj_feature=obj_feature.numpy(), link=blink[0].astype(np.int64),
             X=X, Y=Y, region_feature=region_feature.numpy(), softmax_obj_vec=soft.numpy(),
             out_s_map=s_map.numpy(), out_o_map=o_map.numpy(), out_v_map=v_map.numpy(),
             out_lca=np.asarray(lca_list[0]).astype(np.int64), out_sem=sem.numpy())
    np.savez_compressed(os.path.join(OUT, "py_pair_head.npz"), **p)
    # ------------------------------------------------------------------ detected-box branch
    # prepare_det_rois -> get_det_boxes -> box_results_with_nms_and_limit (model_builder_rel.py:1212-1340), the
    # reference's Python as is.  Its NMS call (utils/boxes.py:324-328 -> cython_nms, not buildable here) is served
    # by the reference's own compiled csrc/cpu/nms_cpu.cpp - the same greedy algorithm (">=", "+1" areas,
    # ascending-index result); scores are distinct, so the two orderings (argsort()[::-1] / sort(descending)) agree.
    ref_c = sys.modules["model._C"]

    def nms_via_ref_cpu(dets, thresh):
        if dets.shape[0] == 0:
            return []
        return ref_c.nms(torch.from_numpy(np.ascontiguousarray(dets[:, :4])),
                         torch.from_numpy(np.ascontiguousarray(dets[:, 4])), float(thresh)).numpy()
    sys.modules["utils.cython_nms"].nms = nms_via_ref_cpu
    cfg.TEST.NMS = 0.5
    cfg.TEST.SCORE_THRESH = 0.05
    cfg.TEST.DETECTIONS_PER_IM = 100
    cfg.TEST.BBOX_REG = True
    cfg.TEST.SOFT_NMS.ENABLED = False
    cfg.TEST.BBOX_VOTE.ENABLED = False
    cfg.MODEL.CLS_AGNOSTIC_BBOX_REG = False
    cfg.ENABLE_FRAME_PRE_PROCESSING = False
    d = {}
    dn = types.SimpleNamespace()
    for meth in ("get_det_boxes", "box_results_with_nms_and_limit"):
        setattr(dn, meth, types.MethodType(getattr(Generalized_RCNN, meth), dn))
    dcases = []
    for name, R, ncls, sharp in (("r300", 300, 9, 3.0), ("r40", 40, 9, 1.0), ("r1", 1, 9, 4.0)):
        cfg.MODEL.NUM_CLASSES = ncls
        im_info = torch.tensor([[600.0, 800.0, 1.6]])
        rb = synth_boxes(rng, R, 5.0, 790.0)
        rb[:, 3] = np.minimum(rb[:, 3], 590.0)
        rois_d = np.concatenate([np.zeros((R, 1), np.float32), rb], 1)
        cls_scores = torch.softmax(torch.randn((R, ncls), generator=gen) * sharp, 1)
        bbox_pred = torch.randn((R, 4 * ncls), generator=gen) * 0.5
        dr, dl, dsc, dmap = Generalized_RCNN.prepare_det_rois(dn, rois_d, cls_scores, bbox_pred, im_info, "vidvrd_val",
                                                              score_thresh=cfg.TEST.SCORE_THRESH, batch_size=1)
        dcases.append(name)
        d[name + "_rois"], d[name + "_cls_scores"], d[name + "_bbox_pred"] = rois_d, cls_scores.numpy(), bbox_pred.numpy()
        d[name + "_im_info"] = im_info.numpy()
        d[name + "_out_det_rois"], d[name + "_out_det_labels"] = dr, dl
        d[name + "_out_det_scores"], d[name + "_out_inds"] = dsc, np.asarray(dmap).astype(np.int64)
    d["cases"] = jbytes(dcases)
    d["cfg"] = jbytes(dict(nms=0.5, score_thresh=0.05, detections_per_im=100, reg_weights=list(cfg.MODEL.BBOX_REG_WEIGHTS)))
    np.savez_compressed(os.path.join(OUT, "py_det_branch.npz"), **d)

    for f in sorted(os.listdir(OUT)):
        print("%-28s %8d bytes" % (f, os.path.getsize(os.path.join(OUT, f))))


if __name__ == "__main__":
    main()
