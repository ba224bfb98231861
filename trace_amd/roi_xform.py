"""`roi_feature_transform`: both copies of the reference (lib/modeling_rel/model_builder_rel.py:1342-1396,
lib/combine_tree/cluster_tree.py:689-734) behind one function - FPN-aware dispatch (per-level ROIAlign / ROIPool,
concatenation, un-shuffle by `<blob>_idx_restore_int32`) or the single-level path - on the HIP ROI kernels.

Layouts: a float32 (B,C,H,W) blob goes through the `_C` drop-in kernels and returns (K,C,res,res) float32, bit-identical
to the reference's CPU kernel; a channels-last bf16 (B,H,W,C) blob (what the fused pipeline keeps) goes through
`trace_roi_align_fwd_nhwc_bf16` and returns (K,res,res,C) bf16.
"""
import numpy as np
import torch

from . import ops
from .roi_layers import ROIAlign, ROIPool


def _is_nhwc_bf16(x):
    return x.dtype == torch.bfloat16


def _one_level(blob, rois_np, method, resolution, spatial_scale, sampling_ratio):
    rois = torch.from_numpy(np.ascontiguousarray(rois_np, np.float32)).to(blob.device)   # :1389, cluster_tree.py:730
    if _is_nhwc_bf16(blob):
        if method != 'RoIAlign':
            raise RuntimeError("roi_feature_transform: channels-last bf16 blobs support RoIAlign only")
        return ops.roi_align_nhwc(blob.contiguous(), rois, float(spatial_scale), int(resolution), int(sampling_ratio))
    if method == 'RoIPoolF':
        return ROIPool((resolution, resolution), spatial_scale)(blob, rois)
    return ROIAlign((resolution, resolution), spatial_scale, sampling_ratio)(blob, rois)


def roi_feature_transform(blobs_in, rpn_ret, blob_rois='rois', method='RoIAlign', resolution=7,
                          spatial_scale=1. / 16., sampling_ratio=0, k_min=2, k_max=5):
    """Reference signature (+ the FPN level range the reference reads from cfg.FPN.ROI_{MIN,MAX}_LEVEL)."""
    assert method in {'RoIPoolF', 'RoIAlign'}, 'Unknown pooling method: {}'.format(method)
    if isinstance(blobs_in, (list, tuple)):
        # FPN: blobs_in and spatial_scale are in reversed (coarsest first) order (:1361-1363)
        assert len(blobs_in) == k_max - k_min + 1
        outs = []
        for lvl in range(k_min, k_max + 1):
            bl_in = blobs_in[k_max - lvl]
            sc = spatial_scale[k_max - lvl]
            bl_rois = rpn_ret[blob_rois + '_fpn' + str(lvl)]
            if len(bl_rois):
                outs.append(_one_level(bl_in, bl_rois, method, resolution, sc, sampling_ratio))
        shuffled = torch.cat(outs, dim=0)
        restore = torch.from_numpy(np.asarray(rpn_ret[blob_rois + '_idx_restore_int32']).astype('int64', copy=False))
        return shuffled[restore.to(shuffled.device)]
    return _one_level(blobs_in, rpn_ret[blob_rois], method, resolution, spatial_scale, sampling_ratio)


def add_rel_multilevel_rois(blobs, blob_name='rois', k_min=2, k_max=5):
    """`_add_rel_multilevel_rois` (model_builder_rel.py / cluster_tree.py: per-level roi blobs + restore index for one
    roi list) on the host, with utils/fpn.py:11-58's float32 arithmetic (level assignment is a bit-exact contract:
    the device kernel trace_fpn_levels is tested against the same golden)."""
    rois = np.asarray(blobs[blob_name], np.float32)
    lvls = map_rois_to_fpn_levels_np(rois[:, 1:5], k_min, k_max)
    idx_all = []
    for lvl in range(k_min, k_max + 1):
        idx = np.where(lvls == lvl)[0]
        blobs[blob_name + '_fpn' + str(lvl)] = rois[idx, :]
        idx_all.append(idx)
    order = np.concatenate(idx_all) if idx_all else np.zeros(0, np.int64)
    blobs[blob_name + '_idx_restore_int32'] = np.argsort(order).astype(np.int32, copy=False)
    return blobs


def map_rois_to_fpn_levels_np(boxes, k_min, k_max, s0=224, lvl0=4):
    """utils/fpn.py:11-28 with utils/boxes.py:58-73 (host numpy, float32 as the reference)."""
    w = boxes[:, 2] - boxes[:, 0] + 1
    h = boxes[:, 3] - boxes[:, 1] + 1
    areas = w * h
    areas[np.where(areas < 0)[0]] = 0
    s = np.sqrt(areas)
    target = np.floor(lvl0 + np.log2(s / s0 + 1e-6))
    return np.clip(target, k_min, k_max)
