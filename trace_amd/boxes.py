"""Device versions of the reference's numpy/Cython box helpers.

  nms(dets (N,5), thresh)        utils/boxes.py:324-328 -> cython_nms.pyx:37-87
  bbox_overlaps                  utils/cython_bbox.pyx:32-73
  bbox_pair_overlaps             lib/utils_rel/cython_bbox_rel.pyx:37-77
  map_rois_to_fpn_levels         utils/fpn.py:11-28
  add_multilevel_roi_blobs       utils/fpn.py:31-58
  relpn_pairs                    lib/modeling_rel/relpn_heads.py:79-84,117
"""
import torch

from . import _C
from . import _lib as L


def nms(dets, thresh):
    """dets (N,5) [x1,y1,x2,y2,score] f32 on GPU -> kept indices (ascending)."""
    if dets.shape[0] == 0:
        return torch.empty((0,), dtype=torch.int64, device=dets.device)
    L.require_gpu(dets)
    return _C.nms(dets[:, :4].contiguous(), dets[:, 4].contiguous(), thresh, _C.NMS_GE)


def bbox_overlaps(boxes, query_boxes):
    L.require_gpu(boxes, query_boxes)
    boxes = boxes.float().contiguous()
    query_boxes = query_boxes.float().contiguous()
    N, K = boxes.shape[0], query_boxes.shape[0]
    out = torch.empty((N, K), dtype=torch.float32, device=boxes.device)
    with torch.cuda.device(boxes.device):
        L.check(L.lib().trace_bbox_overlaps(L.ptr(boxes), N, L.ptr(query_boxes), K, L.ptr(out),
                                            L.stream()), "bbox_overlaps")
    return out


def bbox_pair_overlaps(boxes1, boxes2):
    L.require_gpu(boxes1, boxes2)
    assert boxes1.shape[0] == boxes2.shape[0]
    boxes1 = boxes1.float().contiguous()
    boxes2 = boxes2.float().contiguous()
    N = boxes1.shape[0]
    out = torch.empty((N,), dtype=torch.float32, device=boxes1.device)
    with torch.cuda.device(boxes1.device):
        L.check(L.lib().trace_bbox_pair_overlaps(L.ptr(boxes1), L.ptr(boxes2), N, L.ptr(out),
                                                 L.stream()), "bbox_pair_overlaps")
    return out


def map_rois_to_fpn_levels(rois, k_min, k_max, canonical_scale=224, canonical_level=4):
    """rois (N,4) f32 on GPU -> target levels (N,) f32 (fpn.py:11-28)."""
    L.require_gpu(rois)
    rois = rois.float().contiguous()
    N = rois.shape[0]
    lv = torch.empty((N,), dtype=torch.float32, device=rois.device)
    with torch.cuda.device(rois.device):
        L.check(L.lib().trace_fpn_levels(L.ptr(rois), N, int(k_min), int(k_max),
                                         float(canonical_scale), float(canonical_level), L.ptr(lv),
                                         L.stream()), "fpn_levels")
    return lv


def add_multilevel_roi_blobs(blobs, blob_prefix, rois, target_lvls, lvl_min, lvl_max):
    """fpn.py:31-58 on device tensors: per-level roi lists + int32 restore index
    (stable grouping by level, then argsort of the concatenated index order)."""
    order = []
    for lvl in range(lvl_min, lvl_max + 1):
        idx = torch.nonzero(target_lvls == lvl).flatten()
        blobs[blob_prefix + "_fpn" + str(lvl)] = rois[idx]
        order.append(idx)
    order = torch.cat(order) if order else torch.empty((0,), dtype=torch.int64, device=rois.device)
    blobs[blob_prefix + "_idx_restore_int32"] = torch.argsort(order, stable=True).to(torch.int32)
    return blobs


def relpn_pairs(det_rois):
    """All ordered (sbj,obj) pairs of the n rois, self pairs removed, plus union
    rois (relpn_heads.py:79-84,39-46,117; boxes_rel.py:100-106)."""
    L.require_gpu(det_rois)
    det_rois = det_rois.float().contiguous()
    n = det_rois.shape[0]
    P = n * (n - 1) if n > 1 else n
    dev = det_rois.device
    sbj = torch.empty((P,), dtype=torch.int32, device=dev)
    obj = torch.empty((P,), dtype=torch.int32, device=dev)
    rel = torch.empty((P, 5), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        L.check(L.lib().trace_relpn_pairs(L.ptr(det_rois), n, L.ptr(sbj), L.ptr(obj), L.ptr(rel),
                                          L.stream()), "relpn_pairs")
    return sbj, obj, rel


# ---------------------------------------------------------------------------------------------------
# Detected-box branch: lib/modeling_rel/model_builder_rel.py:1212-1340
# ---------------------------------------------------------------------------------------------------
BBOX_XFORM_CLIP = 4.135166556742356  # np.log(1000. / 16.), lib/core/config.py:1081


def nms_grouped(boxes, scores, group, thresh, mode=_C.NMS_GE):
    """Per-group (per-class) greedy NMS in one pass: boxes (N,4), scores (N,), group (N,) int32 on the
    GPU -> kept original indices, ascending.  Same keeps as running `nms` on every group separately."""
    L.require_gpu(boxes, scores, group)
    N = boxes.shape[0]
    if N == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    boxes = boxes.float().contiguous()
    scores = scores.float().contiguous()
    group = group.to(torch.int32).contiguous()
    lib = L.lib()
    ws_bytes = lib.trace_nms_workspace_bytes(N)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=boxes.device)
    keep = torch.empty((N,), dtype=torch.int64, device=boxes.device)
    num = torch.zeros((1,), dtype=torch.int32, device=boxes.device)
    with torch.cuda.device(boxes.device):
        L.check(lib.trace_nms_grouped(L.ptr(boxes), L.ptr(scores), L.ptr(group), N, float(thresh), int(mode),
                                      L.ptr(keep), L.ptr(num), L.ptr(ws), ws_bytes, L.stream()), "nms_grouped")
    return keep[: int(num.item())]


def bbox_transform_clip(boxes, deltas, weights, im_hw, xform_clip=BBOX_XFORM_CLIP):
    """utils/boxes.py:160-200 + :142-157: boxes (R,4), deltas (R,4K) f32 on the GPU -> decoded, clipped (R,4K)."""
    L.require_gpu(boxes, deltas)
    boxes = boxes.float().contiguous()
    deltas = deltas.float().contiguous()
    R = boxes.shape[0]
    if deltas.shape[0] != R or deltas.shape[1] % 4:
        raise RuntimeError("bbox_transform: deltas must be (R, 4K)")
    K = deltas.shape[1] // 4
    out = torch.empty_like(deltas)
    if R == 0:
        return out
    wx, wy, ww, wh = (float(w) for w in weights)
    with torch.cuda.device(boxes.device):
        L.check(L.lib().trace_bbox_transform_clip(L.ptr(boxes), L.ptr(deltas), R, K, wx, wy, ww, wh,
                                                  float(xform_clip), float(im_hw[0]), float(im_hw[1]),
                                                  L.ptr(out), L.stream()), "bbox_transform_clip")
    return out


def get_det_boxes(boxes, scores, box_deltas, h_and_w, bbox_reg=True, cls_agnostic=False,
                  reg_weights=(10., 10., 5., 5.)):
    """model_builder_rel.py:1261-1278: (scores, per-class boxes (R, 4*C))."""
    C = scores.shape[1]
    if bbox_reg:
        if cls_agnostic:
            box_deltas = box_deltas[:, -4:]
        pred = bbox_transform_clip(boxes, box_deltas, reg_weights, h_and_w)
        if cls_agnostic:
            pred = pred.repeat(1, C)
    else:
        pred = boxes.float().repeat(1, C)
    return scores, pred


def box_results_with_nms_and_limit(scores, boxes, score_thresh=0.05, nms_thresh=0.3, detections_per_im=100):
    """model_builder_rel.py:1280-1340 for one image, on the GPU: score threshold -> per-class NMS (one
    grouped pass; `>=` Cython semantics) -> keep the `detections_per_im` best over all classes.
    scores (R,C), boxes (R,4C).  Returns scores, boxes (n,4), labels (f32), im_inds (roi index of each
    detection), ordered by class then roi index as the reference."""
    L.require_gpu(scores, boxes)
    R, C = scores.shape
    sc = scores.float()
    cand = torch.nonzero(sc[:, 1:].t() > score_thresh)            # (class-1, roi) pairs, class-major order
    cls = cand[:, 0] + 1
    roi = cand[:, 1]
    s_j = sc[roi, cls]
    b_j = boxes.float().view(R, C, 4)[roi, cls]
    keep = nms_grouped(b_j, s_j, cls, nms_thresh)
    s_k, b_k, c_k, r_k = s_j[keep], b_j[keep], cls[keep], roi[keep]
    if detections_per_im > 0 and s_k.numel() > detections_per_im:
        image_thresh = torch.sort(s_k).values[-detections_per_im]
        sel = s_k >= image_thresh
        s_k, b_k, c_k, r_k = s_k[sel], b_k[sel], c_k[sel], r_k[sel]
    return s_k, b_k, c_k.float(), r_k


def prepare_det_rois(rois, cls_scores, bbox_pred, im_info, score_thresh=0.05, nms_thresh=0.3,
                     detections_per_im=100, batch_size=1, pre_boxes=None, **det_kw):
    """model_builder_rel.py:1212-1259: rois (R,5) [batch,x1,y1,x2,y2] f32 GPU, cls_scores (R,C),
    bbox_pred (R,4C), im_info (B,3) [H,W,scale] (host).  Returns det_rois (n,5), det_labels,
    det_scores, det_inds_mapping (device tensors)."""
    im_info = torch.as_tensor(im_info, dtype=torch.float32).cpu()
    out_r, out_l, out_s, out_i = [], [], [], []
    for im_i in range(batch_size):
        inds = torch.nonzero((rois[:, 0] - im_i).abs() < 1e-06).flatten()
        scale = float(im_info[im_i, 2])
        im_boxes = rois[inds, 1:5] / scale
        im_scores = cls_scores[inds].reshape(inds.numel(), -1)
        im_deltas = bbox_pred[inds].reshape(inds.numel(), -1)
        if pre_boxes is not None:                       # ENABLE_FRAME_PRE_PROCESSING (:1241-1242)
            im_boxes_c = pre_boxes[inds, 1:5].repeat(1, im_scores.shape[1])
        else:
            hw = (float(im_info[im_i, 0]) / scale, float(im_info[im_i, 1]) / scale)
            im_scores, im_boxes_c = get_det_boxes(im_boxes, im_scores, im_deltas, hw, **det_kw)
        s, b, l, ii = box_results_with_nms_and_limit(im_scores, im_boxes_c, score_thresh, nms_thresh, detections_per_im)
        out_r.append(torch.cat([torch.full((b.shape[0], 1), float(im_i), device=b.device), b * scale], 1))
        out_l.append(l)
        out_s.append(s)
        out_i.append(inds[ii])
    return torch.cat(out_r), torch.cat(out_l), torch.cat(out_s), torch.cat(out_i)


def rois_union_np(rois1, rois2):
    """lib/utils_rel/boxes_rel.py:100-106 on the host (the pair lists are built there): (P,5) x (P,5) -> (P,5)."""
    import numpy as np
    assert (rois1[:, 0] == rois2[:, 0]).all()
    xmin = np.minimum(rois1[:, 1], rois2[:, 1])
    ymin = np.minimum(rois1[:, 2], rois2[:, 2])
    xmax = np.maximum(rois1[:, 3], rois2[:, 3])
    ymax = np.maximum(rois1[:, 4], rois2[:, 4])
    return np.vstack((rois1[:, 0], xmin, ymin, xmax, ymax)).transpose()


def relpn_pair_indices(n, det_labels=None, dataset="vidvrd"):
    """Host-side pair enumeration of single_scale_relpn_outputs.forward (relpn_heads.py:71-84,39-55):
    all ordered pairs without self pairs; on Action Genome ('ag' in the dataset name) only persons
    (label <= 1) are subjects.  Returns numpy int64 (sbj_inds, obj_inds)."""
    import numpy as np
    person = np.where(np.asarray(det_labels) <= 1)[0] if det_labels is not None else np.zeros(0, np.int64)
    if dataset.find("ag") >= 0 and len(person) > 0:
        sbj = np.repeat(person, n)
        obj = np.tile(np.arange(n), person.shape[0])
        if n > 1:
            mask = np.ones(sbj.shape[0], dtype=bool)
            mask[person + n * np.arange(person.shape[0])] = False
            sbj, obj = sbj[mask], obj[mask]
    else:
        sbj = np.repeat(np.arange(n), n)
        obj = np.tile(np.arange(n), n)
        if n > 1:
            mask = np.ones(sbj.shape[0], dtype=bool)
            mask[np.arange(n) * (n + 1)] = False
            sbj, obj = sbj[mask], obj[mask]
    return sbj.astype(np.int64), obj.astype(np.int64)


def ovlp_filter(sbj_boxes, obj_boxes):
    """relpn_heads.py:94-105 USE_OVLP_FILTER: indices of the pairs whose boxes overlap (IoU > 0, "+1"
    areas, f32 as cython_bbox_rel.pyx:37-77); every pair when none overlaps.  Host numpy in/out (the
    pair lists are built on the host); the IoU itself is trace_bbox_pair_overlaps' expression."""
    import numpy as np
    a = np.asarray(sbj_boxes, np.float32)
    b = np.asarray(obj_boxes, np.float32)
    one = np.float32(1)
    iw = np.minimum(a[:, 2], b[:, 2]) - np.maximum(a[:, 0], b[:, 0]) + one
    ih = np.minimum(a[:, 3], b[:, 3]) - np.maximum(a[:, 1], b[:, 1]) + one
    inds = np.where((iw > 0) & (ih > 0))[0]       # IoU > 0  <=>  positive intersection (areas are positive)
    return inds if inds.size > 0 else np.arange(a.shape[0])
