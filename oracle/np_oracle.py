"""TEST INFRASTRUCTURE ONLY — numpy restatements of the reference's numpy glue.

Pinned by execution: tests/golden/py_box_glue.npz, py_tree.npz and py_relpn.npz hold outputs of the
reference's own functions (run on the CPU of the build container by tools/make_goldens_py.py) and
tests/test_oracle_goldens_py.py checks every function here against them bit for bit.  bbox_overlaps /
bbox_pair_overlaps (oracle/trace_oracle.c) are pinned since round 3 by the reference's own Cython modules compiled
here (tests/golden/cy_overlaps.npz, tests/test_oracle_goldens_r3.py); box_results_with_nms_and_limit is pinned by
tests/golden/py_det_branch.npz.
"""
import math

import numpy as np


def boxes_area(boxes):
    """Detectron_pytorch/lib/utils/boxes.py:58-73."""
    w = (boxes[:, 2] - boxes[:, 0] + 1)
    h = (boxes[:, 3] - boxes[:, 1] + 1)
    areas = w * h
    neg_area_idx = np.where(areas < 0)[0]
    return areas, neg_area_idx


def map_rois_to_fpn_levels(rois, k_min, k_max, s0=224, lvl0=4):
    """Detectron_pytorch/lib/utils/fpn.py:11-28 (cfg.FPN.ROI_CANONICAL_SCALE=224,
    ROI_CANONICAL_LEVEL=4, lib/core/config.py)."""
    areas, neg_idx = boxes_area(rois)
    areas[neg_idx] = 0
    s = np.sqrt(areas)
    target_lvls = np.floor(lvl0 + np.log2(s / s0 + 1e-6))
    target_lvls = np.clip(target_lvls, k_min, k_max)
    return target_lvls


def add_multilevel_roi_blobs(blobs, blob_prefix, rois, target_lvls, lvl_min, lvl_max):
    """Detectron_pytorch/lib/utils/fpn.py:31-58."""
    rois_idx_order = np.empty((0,))
    rois_stacked = np.zeros((0, 5), dtype=np.float32)
    for lvl in range(lvl_min, lvl_max + 1):
        idx_lvl = np.where(target_lvls == lvl)[0]
        blobs[blob_prefix + '_fpn' + str(lvl)] = rois[idx_lvl, :]
        rois_idx_order = np.concatenate((rois_idx_order, idx_lvl))
        rois_stacked = np.vstack([rois_stacked, blobs[blob_prefix + '_fpn' + str(lvl)]])
    rois_idx_restore = np.argsort(rois_idx_order).astype(np.int32, copy=False)
    blobs[blob_prefix + '_idx_restore_int32'] = rois_idx_restore
    assert (rois_stacked[rois_idx_restore] == rois).all()
    return blobs


def bbox_transform(boxes, deltas, weights=(1.0, 1.0, 1.0, 1.0), xform_clip=math.log(1000. / 16.)):
    """Detectron_pytorch/lib/utils/boxes.py:160-200 (cfg.BBOX_XFORM_CLIP = log(1000/16))."""
    if boxes.shape[0] == 0:
        return np.zeros((0, deltas.shape[1]), dtype=deltas.dtype)
    boxes = boxes.astype(deltas.dtype, copy=False)
    widths = boxes[:, 2] - boxes[:, 0] + 1.0
    heights = boxes[:, 3] - boxes[:, 1] + 1.0
    ctr_x = boxes[:, 0] + 0.5 * widths
    ctr_y = boxes[:, 1] + 0.5 * heights
    wx, wy, ww, wh = weights
    dx = deltas[:, 0::4] / wx
    dy = deltas[:, 1::4] / wy
    dw = deltas[:, 2::4] / ww
    dh = deltas[:, 3::4] / wh
    dw = np.minimum(dw, xform_clip)
    dh = np.minimum(dh, xform_clip)
    pred_ctr_x = dx * widths[:, np.newaxis] + ctr_x[:, np.newaxis]
    pred_ctr_y = dy * heights[:, np.newaxis] + ctr_y[:, np.newaxis]
    pred_w = np.exp(dw) * widths[:, np.newaxis]
    pred_h = np.exp(dh) * heights[:, np.newaxis]
    pred_boxes = np.zeros(deltas.shape, dtype=deltas.dtype)
    pred_boxes[:, 0::4] = pred_ctr_x - 0.5 * pred_w
    pred_boxes[:, 1::4] = pred_ctr_y - 0.5 * pred_h
    pred_boxes[:, 2::4] = pred_ctr_x + 0.5 * pred_w - 1
    pred_boxes[:, 3::4] = pred_ctr_y + 0.5 * pred_h - 1
    return pred_boxes


def clip_tiled_boxes(boxes, im_shape):
    """Detectron_pytorch/lib/utils/boxes.py:142-157."""
    boxes[:, 0::4] = np.maximum(np.minimum(boxes[:, 0::4], im_shape[1] - 1), 0)
    boxes[:, 1::4] = np.maximum(np.minimum(boxes[:, 1::4], im_shape[0] - 1), 0)
    boxes[:, 2::4] = np.maximum(np.minimum(boxes[:, 2::4], im_shape[1] - 1), 0)
    boxes[:, 3::4] = np.maximum(np.minimum(boxes[:, 3::4], im_shape[0] - 1), 0)
    return boxes


def argsort_desc(scores):
    """lib/datasets_rel/pytorch_misc.py:28-35."""
    return np.column_stack(np.unravel_index(np.argsort(-scores.ravel()), scores.shape))


def rois_union(rois1, rois2):
    """lib/utils_rel/boxes_rel.py:100-106."""
    assert (rois1[:, 0] == rois2[:, 0]).all()
    xmin = np.minimum(rois1[:, 1], rois2[:, 1])
    ymin = np.minimum(rois1[:, 2], rois2[:, 2])
    xmax = np.maximum(rois1[:, 3], rois2[:, 3])
    ymax = np.maximum(rois1[:, 4], rois2[:, 4])
    return np.vstack((rois1[:, 0], xmin, ymin, xmax, ymax)).transpose()


def relpn_pairs(det_rois, det_labels=None, dataset_name="vidvrd"):
    """lib/modeling_rel/relpn_heads.py:71-88,117 (+ remove_self_pairs :39-46,
    sbj_adapt_remove_self_pairs :48-55: on Action Genome only persons, label <= 1, are subjects)."""
    n = det_rois.shape[0]
    person_inds = np.where(det_labels <= 1)[0] if det_labels is not None else np.zeros((0,), np.int64)
    if dataset_name.find('ag') >= 0 and len(person_inds) > 0:
        sbj_inds = np.repeat(person_inds, n)
        obj_inds = np.tile(np.arange(n), person_inds.shape[0])
        if n > 1:
            mask = np.ones(sbj_inds.shape[0], dtype=bool)
            for i in range(person_inds.shape[0]):
                mask[person_inds[i] + n * i] = False
            keeps = np.where(mask)[0]
            sbj_inds = sbj_inds[keeps]
            obj_inds = obj_inds[keeps]
    else:
        sbj_inds = np.repeat(np.arange(n), n)
        obj_inds = np.tile(np.arange(n), n)
        if n > 1:
            mask = np.ones(sbj_inds.shape[0], dtype=bool)
            for i in range(n):
                mask[i + n * i] = False
            keeps = np.where(mask)[0]
            sbj_inds = sbj_inds[keeps]
            obj_inds = obj_inds[keeps]
    sbj_rois = det_rois[sbj_inds]
    obj_rois = det_rois[obj_inds]
    rel_rois = rois_union(sbj_rois, obj_rois)
    return sbj_inds, obj_inds, rel_rois


# ---------------------------------------------------------------------------
# Hierarchical box tree: lib/combine_tree/cluster_tree.py:504-649, 400-430
# ---------------------------------------------------------------------------
def gt_box_pairs(sbj_gt_boxes, obj_gt_boxes, im_scale, sbj_gt_classes=None, obj_gt_classes=None):
    """The gt-box branch's host arithmetic, lib/modeling_rel/model_builder_rel.py:773-810 (+ :845-858 when the gt
    classes are given = use_gt_labels): annotated pairs de-duplicated (np.unique over the int32 8-tuples), boxes
    de-duplicated into det_rois, pair -> box indices.  Pinned by tests/golden/py_gt_branch.npz (those very lines of
    the reference, executed on the same inputs by tools/make_goldens_r2.py)."""
    sbj_boxes = sbj_gt_boxes.copy()                                                  # :776-777
    obj_boxes = obj_gt_boxes.copy()
    pair_gt_boxes = np.concatenate((sbj_boxes, obj_boxes), axis=1).astype(np.int32)  # :780
    pair_gt_boxes, pair_gt_index = np.unique(pair_gt_boxes, return_index=True, axis=0)   # :781
    sbj_boxes = pair_gt_boxes[:, :4].astype(np.float32)
    obj_boxes = pair_gt_boxes[:, 4:].astype(np.float32)
    all_boxes = np.concatenate((sbj_boxes, obj_boxes)).astype(np.int32)              # :787
    unique_all_boxes, gt_boxes_id = np.unique(all_boxes, return_inverse=True, axis=0)    # :788
    gt_boxes_id = np.asarray(gt_boxes_id).reshape(-1)
    sbj_gt_boxes_id = gt_boxes_id[:sbj_boxes.shape[0]]
    obj_gt_boxes_id = gt_boxes_id[sbj_boxes.shape[0]:]
    unique_all_boxes = unique_all_boxes.astype(np.float32)
    im_scale = np.float32(im_scale)
    sbj_rois = sbj_boxes * im_scale                                                  # :794-796
    obj_rois = obj_boxes * im_scale
    all_rois = unique_all_boxes * im_scale
    rep = 0 * np.ones((sbj_rois.shape[0], 1), dtype=np.float32)                      # blob_utils.ones: float32 (:797-798)
    rep_det = 0 * np.ones((all_rois.shape[0], 1), dtype=np.float32)
    sbj_rois = np.hstack((rep, sbj_rois))
    obj_rois = np.hstack((rep, obj_rois))
    all_rois = np.hstack((rep_det, all_rois))
    out = dict(sbj_rois=sbj_rois, obj_rois=obj_rois, rel_rois=rois_union(sbj_rois, obj_rois), det_rois=all_rois,
               sbj_inds=sbj_gt_boxes_id, obj_inds=obj_gt_boxes_id, pair_gt_index=pair_gt_index)
    if sbj_gt_classes is not None:                                                   # :845-858
        sbj_labels = sbj_gt_classes[pair_gt_index]
        obj_labels = obj_gt_classes[pair_gt_index]
        all_labels = np.concatenate((sbj_labels, obj_labels))
        _, gt_labels_id = np.unique(gt_boxes_id, return_index=True)
        det_labels = all_labels[gt_labels_id]
        out.update(sbj_labels=sbj_labels, obj_labels=obj_labels, det_labels=det_labels,
                   sbj_scores=np.ones_like(sbj_labels, dtype=np.float32),
                   obj_scores=np.ones_like(obj_labels, dtype=np.float32),
                   det_scores=np.ones_like(det_labels, dtype=np.float32))
    return out


def _get_distance(x, y):
    return ((x - y) ** 2).sum(-1)  # cluster_tree.py:610-611 ('euclid')


def _get_cluster(ids, bbox_feat):
    """cluster_tree.py:595-607."""
    X, Y = np.meshgrid(ids, ids, indexing='ij')
    distance = _get_distance(bbox_feat[X], bbox_feat[Y])
    distance = distance.reshape(len(ids), -1)
    score = -np.sum(np.exp(-distance), axis=1)
    id_id = np.argsort(score)
    ids = ids[id_id]
    c_id = [i for i in range(0, len(ids), 2)]
    not_c_id = [i for i in range(1, len(ids), 2)]
    return np.hstack((ids[c_id], ids[not_c_id]))


def _cluster_algorithm(bbox_feat, cluster_num, l, r):
    """cluster_tree.py:558-593."""
    ids = np.arange(l, r + 1)
    ids = _get_cluster(ids, bbox_feat)
    cluster_id = ids[:cluster_num]
    combine_id = ids[cluster_num:]
    combine_flat, cluster_flat = np.meshgrid(combine_id, cluster_id)
    combine_flat = combine_flat.reshape(-1)
    cluster_flat = cluster_flat.reshape(-1)
    distance = _get_distance(bbox_feat[cluster_flat], bbox_feat[combine_flat])
    distance = distance.reshape(cluster_num, -1)
    depend_id = np.argmin(distance, axis=0) if distance.shape[1] else np.zeros((0,), np.int64)
    new_node = {}
    for i in range(cluster_num):
        new_node[r + 1 + i] = [cluster_id[i]]
    combine_len = r - l + 1 - cluster_num
    for i in range(combine_len):
        new_node[r + 1 + depend_id[i]].append(combine_id[i])
    return new_node


def _get_new_bbox_feat(ori, new_node_dict):
    """cluster_tree.py:617-628 ('coordinate')."""
    new_node_feat = []
    for head, son_list in new_node_dict.items():
        son_feat = ori[son_list]
        new_node_feat.append(np.array([son_feat[:, 0].min(), son_feat[:, 1].min(),
                                       son_feat[:, 2].max(), son_feat[:, 3].max()]))
    new_node_feat = np.stack(new_node_feat, axis=0)
    return np.vstack((ori, new_node_feat))


def node_cluster(leaf_boxes, max_tree_height=7, partition=2):
    """cluster_tree.py:504-556 for ONE image (batch handled by the caller).
    leaf_boxes (n,4) -> (layer list of {parent: [children]}, tree boxes (n+m,4))."""
    n = leaf_boxes.shape[0]
    layers = []
    cluster_num = int(math.ceil(n))
    bbox_feat = leaf_boxes.copy()
    l, r = 0, cluster_num - 1
    for _ in range(max_tree_height):
        cluster_num = int(math.ceil(float(cluster_num) / partition))
        new_node = _cluster_algorithm(bbox_feat, cluster_num, l, r)
        layers.append(new_node)
        bbox_feat = _get_new_bbox_feat(bbox_feat, new_node)
        l = r + 1
        r = r + cluster_num
        if cluster_num <= 1:
            break
    if cluster_num > 1:
        new_node = _cluster_algorithm(bbox_feat, 1, l, r)
        layers.append(new_node)
        bbox_feat = _get_new_bbox_feat(bbox_feat, new_node)
    return layers, bbox_feat


def get_A_and_link(n_tree, layers):
    """cluster_tree.py:400-430 for one image."""
    A_T = np.zeros((n_tree, n_tree), dtype=np.float32)
    A = np.zeros((n_tree, n_tree), dtype=np.float32)
    layer_list = layers[::-1]
    link = (-1) * np.ones((n_tree, len(layer_list) + 1), dtype=np.int64)
    link[n_tree - 1, len(layer_list)] = n_tree - 1
    j = len(layer_list) - 1
    for layer in layer_list:
        for fa, son in layer.items():
            A_T[fa, son] = 1.0
            A[son, fa] = 1.0
            link[son, j + 1:] = link[fa, j + 1:]
            link[son, j] = son
        j = j - 1
    return A, A_T, link


def get_lca(X, Y, link_mat):
    """lib/modeling_rel/model_builder_rel.py:1614-1619."""
    pos = (link_mat[X] != link_mat[Y]) * link_mat[X]
    pos = pos.argmax(1) + 1
    x_link_mat = link_mat[X]
    return x_link_mat[range(len(pos)), pos]


def temporal_nonleaf_rois(layers, n_tree, n_leaf, temporal_boxes):
    """lib/combine_tree/cluster_tree3d.py:368-391 for one image ('coordinate').
    temporal_boxes (n,T,4) -> (m,T,4) hull boxes of the non-leaf nodes."""
    T = temporal_boxes.shape[1]
    new_node_feat = np.zeros((n_tree, T, 4), dtype=np.float32)
    new_node_feat[:n_leaf] = temporal_boxes
    out = []
    for layer in layers:
        for fa, son in layer.items():
            son_feat = new_node_feat[np.array(son)]
            head = np.array([son_feat[:, :, 0].min(0), son_feat[:, :, 1].min(0),
                             son_feat[:, :, 2].max(0), son_feat[:, :, 3].max(0)])
            new_node_feat[fa] = np.transpose(head)
            out.append(new_node_feat[fa])
    return np.stack(out, axis=0)


def topk_triplets(prd_scores, sbj_scores, obj_scores, prd_k, topk):
    """lib/datasets_rel/task_evaluation_vg_and_vrd.py:207-250 ranking core:
    per pair take the prd_k best predicates, score = sbj*obj*prd, global top-k.
    Returns (scores, pair index, predicate index) sorted by descending score."""
    det_scores_prd = prd_scores  # (P, n_prd)
    det_labels_prd = np.argsort(-det_scores_prd, axis=1)
    det_scores_prd = -np.sort(-det_scores_prd, axis=1)
    det_scores_so = sbj_scores * obj_scores
    det_scores_spo = det_scores_so[:, None] * det_scores_prd[:, :prd_k]
    flat = det_scores_spo.ravel()
    order = np.argsort(-flat, kind='stable')[:topk]
    pair = order // prd_k
    slot = order % prd_k
    return flat[order], pair, det_labels_prd[pair, slot]


# ---------------------------------------------------------------------------
# Detected-box branch: lib/modeling_rel/model_builder_rel.py:1212-1340
# (pinned by tests/golden/py_det_branch.npz)
# ---------------------------------------------------------------------------
def box_results_with_nms_and_limit(scores, boxes, num_classes, score_thresh=0.05, nms_thresh=0.5,
                                   detections_per_im=100, nms_fn=None):
    """model_builder_rel.py:1280-1340 (soft-NMS and box voting disabled, config.py:390).
    nms_fn(dets (k,5) f32, thresh) -> kept indices ascending; default = the C oracle's greedy NMS
    (cython_nms.pyx:37-87 semantics)."""
    if nms_fn is None:
        from . import c_oracle
        nms_fn = lambda d, t: c_oracle.nms(np.ascontiguousarray(d[:, :4]), np.ascontiguousarray(d[:, 4]), t, 0)
    cls_boxes = [[] for _ in range(num_classes)]
    im_inds = []
    for j in range(1, num_classes):
        inds = np.where(scores[:, j] > score_thresh)[0]
        scores_j = scores[inds, j]
        boxes_j = boxes[inds, j * 4:(j + 1) * 4]
        dets_j = np.hstack((boxes_j, scores_j[:, np.newaxis])).astype(np.float32, copy=False)
        keep = np.asarray(nms_fn(dets_j, nms_thresh), np.int64) if dets_j.shape[0] else np.zeros((0,), np.int64)
        nms_dets = dets_j[keep, :]
        im_inds.append(inds[keep])
        label_j = np.ones((nms_dets.shape[0], 1), dtype=np.float32) * j
        cls_boxes[j] = np.hstack((nms_dets, label_j))
    if detections_per_im > 0:
        image_scores = np.hstack([cls_boxes[j][:, -2] for j in range(1, num_classes)])
        if len(image_scores) > detections_per_im:
            image_thresh = np.sort(image_scores)[-detections_per_im]
            for j in range(1, num_classes):
                keep = np.where(cls_boxes[j][:, -2] >= image_thresh)[0]
                im_inds[j - 1] = im_inds[j - 1][keep]
                cls_boxes[j] = cls_boxes[j][keep, :]
    im_results = np.vstack([cls_boxes[j] for j in range(1, num_classes)])
    return im_results[:, -2], im_results[:, :-2], im_results[:, -1], np.concatenate(im_inds, 0)


def prepare_det_rois(rois, scores, box_deltas, im_info, batch_size=1, score_thresh=0.05, nms_thresh=0.5,
                     detections_per_im=100, reg_weights=(10., 10., 5., 5.)):
    """model_builder_rel.py:1212-1259 + get_det_boxes :1261-1278 (TEST.BBOX_REG, class-specific deltas)."""
    det_rois = np.empty((0, 5), dtype=np.float32)
    det_labels = np.empty((0), dtype=np.float32)
    det_scores = np.empty((0), dtype=np.float32)
    det_inds_mapping = np.empty((0), dtype=np.int32)
    for im_i in range(batch_size):
        inds = np.where(abs(rois[:, 0] - im_i) < 1e-06)[0]
        im_boxes = rois[inds, 1:5] / im_info[im_i, 2]
        im_scores = scores[inds].reshape([-1, scores.shape[-1]])
        im_box_deltas = box_deltas[inds].reshape([-1, box_deltas.shape[-1]])
        pred = bbox_transform(im_boxes, im_box_deltas, reg_weights)
        pred = clip_tiled_boxes(pred, im_info[im_i][:2] / im_info[im_i][2])
        s, b, l, ii = box_results_with_nms_and_limit(im_scores, pred, im_scores.shape[1], score_thresh, nms_thresh,
                                                     detections_per_im)
        batch_inds = im_i * np.ones((b.shape[0], 1), dtype=np.float32)
        det_rois = np.append(det_rois, np.hstack((batch_inds, b * im_info[im_i, 2])), axis=0)
        det_labels = np.append(det_labels, l, axis=0)
        det_scores = np.append(det_scores, s, axis=0)
        det_inds_mapping = np.append(det_inds_mapping, inds[ii], axis=0)
    return det_rois, det_labels, det_scores, det_inds_mapping


# ---------------------------------------------------------------------------------------------------------------
# RPN proposals
# ---------------------------------------------------------------------------------------------------------------
def generate_anchors(stride=16, sizes=(32, 64, 128, 256, 512), aspect_ratios=(0.5, 1, 2)):
    """Detectron_pytorch/lib/modeling/generate_anchors.py:48-125: (A,4) float64 anchors around a stride x stride cell,
    aspect ratios outer, scales inner."""
    base = float(stride)
    scales = np.asarray(sizes, np.float64) / stride
    ratios = np.asarray(aspect_ratios, np.float64)

    def whctrs(a):
        w, h = a[2] - a[0] + 1, a[3] - a[1] + 1
        return w, h, a[0] + 0.5 * (w - 1), a[1] + 0.5 * (h - 1)

    def mk(ws, hs, xc, yc):
        ws, hs = ws[:, None], hs[:, None]
        return np.hstack((xc - 0.5 * (ws - 1), yc - 0.5 * (hs - 1), xc + 0.5 * (ws - 1), yc + 0.5 * (hs - 1)))

    anchor = np.array([1, 1, base, base], np.float64) - 1
    w, h, xc, yc = whctrs(anchor)
    ws = np.round(np.sqrt(w * h / ratios))
    hs = np.round(ws * ratios)
    ratio_anchors = mk(ws, hs, xc, yc)
    out = []
    for i in range(ratio_anchors.shape[0]):
        w, h, xc, yc = whctrs(ratio_anchors[i])
        out.append(mk(w * scales, h * scales, xc, yc))
    return np.vstack(out)


def generate_proposals(scores, bbox_deltas, im_info, anchors, feat_stride, pre_nms_topN=6000, post_nms_topN=1000,
                       nms_thresh=0.7, min_size=0, nms_fn=None):
    """Detectron_pytorch/lib/modeling/generate_proposals.py:62-200 for one image: scores (A,H,W), bbox_deltas (4A,H,W)
    f32, im_info (3,) [h, w, scale].  Returns (proposals (R,4) f32, scores (R,1) f32, kept positions in the sorted
    top-pre_nms list).  Ties in the score sort: the reference's argsort is numpy's unstable quicksort; this restatement
    orders ties by ascending anchor index (stable), which is what the device path does.
    nms_fn(boxes, scores, thresh) -> ascending kept indices (utils/cython_nms.pyx:37-87 semantics)."""
    A, H, W = scores.shape
    sx = np.arange(0, W) * feat_stride
    sy = np.arange(0, H) * feat_stride
    sx, sy = np.meshgrid(sx, sy, copy=False)
    shifts = np.vstack((sx.ravel(), sy.ravel(), sx.ravel(), sy.ravel())).transpose()
    all_anchors = (anchors[np.newaxis, :, :] + shifts[:, np.newaxis, :]).reshape((H * W * A, 4))
    deltas = bbox_deltas.transpose((1, 2, 0)).reshape((-1, 4))
    sc = scores.transpose((1, 2, 0)).reshape((-1, 1))
    order = np.argsort(-sc.squeeze(1), kind="stable")
    if 0 < pre_nms_topN < len(sc):
        order = order[:pre_nms_topN]
    deltas, all_anchors, sc = deltas[order, :], all_anchors[order, :], sc[order]
    props = clip_tiled_boxes(bbox_transform(all_anchors, deltas, (1.0, 1.0, 1.0, 1.0)), im_info[:2])
    ms = np.float32(min_size) * np.float32(im_info[2])
    ws = props[:, 2] - props[:, 0] + 1
    hs = props[:, 3] - props[:, 1] + 1
    xc = props[:, 0] + ws / 2.
    yc = props[:, 1] + hs / 2.
    keep = np.where((ws >= ms) & (hs >= ms) & (xc < im_info[1]) & (yc < im_info[0]))[0]
    props, sc, pos = props[keep, :], sc[keep], keep
    if nms_thresh > 0:
        k = nms_fn(props, sc.squeeze(1), nms_thresh)
        if post_nms_topN > 0:
            k = k[:post_nms_topN]
        props, sc, pos = props[k, :], sc[k], pos[k]
    return props, sc, pos
