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
