"""Harness I/O around the hot path: which frames make up a clip and how they become the network's
input, and checkpoint loading with the reference's key remapping.

Reference:
  get_frames_blob        lib/core/test_engine_rel.py:420-520 (frame window around the key frame, mean subtraction,
                         resize to cfg.TEMPORAL_SCALES, (1,T,3,h,w) f32 blob, bf_cur_len, f_scale)
  prep_im_for_blob       Detectron_pytorch/lib/utils/blob.py:104-126, get_target_scale :141-148
  load_ckpt_rel          lib/utils_rel/net_rel.py:19-24 (class_coco2ag), :27-92 (make_res101_map), :94-147
                         (make_res50_fpn_map), :149-150 (make_backbone_map), :211-232
  rel_detections.pkl     lib/core/test_rel.py:117-147 (per-frame result dict), lib/core/test_engine_rel.py:308-317 (image
                         name and gt fields), :340-352 (file names per mode, one pickle of the list)
The reference decodes and resizes with OpenCV, which this image does not have: `frame_window` / `clip_from_frames`
take already decoded uint8 arrays from whatever decoder the caller uses, and the bilinear resize is torch's
(align_corners=False, no antialias = cv2.INTER_LINEAR's sampling rule); pixel-level equality with cv2 is unpinned.
"""
import numpy as np
import torch
import torch.nn.functional as F

PIXEL_MEANS = (102.9801, 115.9465, 122.7717)     # lib/core/config.py:1087 (BGR)


def get_target_scale(im_size_min, im_size_max, target_size, max_size):
    """blob.py:141-148."""
    im_scale = float(target_size) / float(im_size_min)
    if np.round(im_scale * im_size_max) > max_size:
        im_scale = float(max_size) / float(im_size_max)
    return im_scale


def frame_window(frame_id, half_number_of_frames, frames_interval, im_id_st=1, exists=lambda j: True):
    """test_engine_rel.py:436-482: the frame ids of the clip around key frame `frame_id` - up to `half` frames before
    it (not below im_id_st), the key frame, up to `half` after it, `frames_interval` apart; the walk stops at the first
    missing file.  Returns (ids, bf_cur_len) with bf_cur_len = number of frames before the key frame."""
    start = frame_id - (half_number_of_frames + 1) * frames_interval
    end = frame_id + (half_number_of_frames + 1) * frames_interval
    ids = []
    for j in range(frame_id, start, -frames_interval):
        if j < im_id_st:
            break
        ids.append(j)
    ids = ids[::-1][:-1]
    ids += list(range(frame_id, end, frames_interval))
    got, bf_cur_len, k = [], 0, 0
    for j in ids:
        if j < im_id_st:
            continue
        if j == frame_id:
            bf_cur_len, k = k, 0
        if not exists(j):
            break
        got.append(j)
        k += 1
    return got, bf_cur_len


def clip_from_frames(frames_u8, target_size, max_size=1000, pixel_means=PIXEL_MEANS, device=None, keep_uint8=False):
    """Decoded BGR uint8 frames (T,H,W,3) (one array or a list of equal-sized frames) -> the reference's blob:
    (1,T,3,h,w) f32 = resize(frame - pixel_means) (prep_im_for_blob: means first, then the bilinear resize to
    `target_size` on the short side, capped at max_size on the long side) and f_scale.
    keep_uint8: when no resize is needed (scale 1) return the frames themselves, (1,T,H,W,3) uint8, for the stem's
    uint8 entry (trace_conv3d_stem16_u8 subtracts the means on the fly: set model.Prd_RCNN.pixel_means)."""
    fr = torch.as_tensor(np.stack(frames_u8) if isinstance(frames_u8, (list, tuple)) else frames_u8)
    if fr.dtype != torch.uint8 or fr.dim() != 4 or fr.shape[3] != 3:
        raise ValueError("clip_from_frames: expected uint8 frames (T,H,W,3)")
    if device is not None:
        fr = fr.to(device)
    T, H, W, _ = fr.shape
    scale = get_target_scale(min(H, W), max(H, W), target_size, max_size)
    if keep_uint8 and scale == 1.0:
        return fr.unsqueeze(0).contiguous(), scale
    x = fr.permute(0, 3, 1, 2).float() - torch.tensor(pixel_means, dtype=torch.float32, device=fr.device).view(1, 3, 1, 1)
    if scale != 1.0:
        # cv2.resize(fx=fy=scale) output size: round(size * scale) (cv::saturate_cast<int> of a double)
        h, w = int(np.round(H * scale)), int(np.round(W * scale))
        x = F.interpolate(x, size=(h, w), mode="bilinear", align_corners=False, antialias=False)
    return x.unsqueeze(0).contiguous(), scale


# ----------------------------------------------------------------------------------------------- checkpoint loading
def class_coco2ag():
    """net_rel.py:19-24: AG class -> COCO class of the detector checkpoint (0-based), and the AG rows that have one."""
    a = np.array([0, 1, -1, 60, -1, 74, -1, -1, 57, -1, -1, 40, -1, -1, -1, -1, -1, -1, -1, 64, -1, -1, -1, -1, 68, -1, -1, 73,
                  49, -1, -1, 58, 61, -1, -1, -1, -1], dtype=np.int64)
    idx = np.where(a >= 0)[0]
    return a[idx] - 1, idx


def _res_block_map(m_name, prefix):
    """Shared tail of the Detectron -> Detectron2 block naming (net_rel.py:34-63 and :110-132).  The block index is the
    ONE character after `resN.` - as in the reference, so blocks 10..22 of res4 map to the names of blocks 1 / 2."""
    parts = m_name.split(".")
    tail = parts[-1]
    i = m_name.find("res") + 3
    s = prefix + m_name[i] + "." + m_name[i + 2] + "."
    return s, parts, tail


def make_res101_map(m_name):
    """net_rel.py:27-92."""
    s = ""
    if m_name.find("Conv_Body.res1.conv1") >= 0:
        s = "backbone.stem.conv1.weight"
    elif m_name.find("Conv_Body.res1.bn") >= 0:
        s = "backbone.stem.conv1.norm." + m_name.split(".")[-1]
    elif m_name.find("Conv_Body.res") >= 0:
        s, parts, tail = _res_block_map(m_name, "backbone.res")
        if m_name.find("conv") >= 0:
            s += "conv" + m_name[m_name.find("conv") + 4] + "." + tail
        elif m_name.find("bn") >= 0:
            s += "conv" + m_name[m_name.find("bn") + 2] + ".norm." + tail
        elif m_name.find("downsample") >= 0:
            d = m_name[m_name.find("downsample") + 11]
            if d == "0":
                s += "shortcut." + tail
            elif d == "1":
                s += "shortcut.norm." + tail
    elif m_name.find("RPN.RPN_conv") >= 0:
        s = "proposal_generator.rpn_head.conv." + m_name.split(".")[-1]
    elif m_name.find("RPN.RPN_cls_score") >= 0:
        s = "proposal_generator.rpn_head.objectness_logits." + m_name.split(".")[-1]
    elif m_name.find("RPN.RPN_bbox_pred") >= 0:
        s = "proposal_generator.rpn_head.anchor_deltas." + m_name.split(".")[-1]
    elif m_name.find("Box_Head.res") >= 0:
        parts = m_name.split(".")
        if m_name.find("bn") >= 0:
            s = "roi_heads.res5." + parts[-3] + ".conv" + m_name[m_name.find("bn") + 2] + ".norm." + parts[-1]
        elif m_name.find("downsample") >= 0:
            d = m_name[m_name.find("downsample") + 11]
            if d == "0":
                s = "roi_heads.res5." + parts[-4] + ".shortcut." + parts[-1]
            elif d == "1":
                s = "roi_heads.res5." + parts[-4] + ".shortcut.norm." + parts[-1]
        else:
            s = "roi_heads.res5." + parts[-3] + "." + parts[-2] + "." + parts[-1]
    elif m_name.find("Box_Head") >= 0:
        parts = m_name.split(".")
        s = "roi_heads.box_head." + parts[-2] + "." + parts[-1]
    elif m_name.find("Box_Outs") >= 0:
        parts = m_name.split(".")
        s = "roi_heads.box_predictor." + parts[-2] + "." + parts[-1]
    return s


def make_res50_fpn_map(m_name):
    """net_rel.py:94-147."""
    s = ""
    parts = m_name.split(".")
    if m_name.find("Conv_Body.conv_top") >= 0:
        s = "backbone.fpn_lateral5." + parts[-1]
    elif m_name.find("Conv_Body.topdown_lateral_modules") >= 0:
        s = "backbone.fpn_lateral" + str(4 - int(parts[-3])) + "." + parts[-1]
    elif m_name.find("Conv_Body.posthoc_modules") >= 0:
        s = "backbone.fpn_output" + str(5 - int(parts[-2])) + "." + parts[-1]
    elif m_name.find("conv_body.res1.conv1") >= 0:
        s = "backbone.bottom_up.stem.conv1.weight"
    elif m_name.find("conv_body.res1.bn") >= 0:
        s = "backbone.bottom_up.stem.conv1.norm." + parts[-1]
    elif m_name.find("conv_body.res") >= 0:
        s, parts, tail = _res_block_map(m_name, "backbone.bottom_up.res")
        if parts[-2].find("conv") >= 0:
            s += "conv" + parts[-2][-1] + "." + tail
        elif m_name.find("bn") >= 0:
            s += "conv" + m_name[m_name.find("bn") + 2] + ".norm." + tail
        elif m_name.find("downsample") >= 0:
            d = m_name[m_name.find("downsample") + 11]
            if d == "0":
                s += "shortcut." + tail
            elif d == "1":
                s += "shortcut.norm." + tail
    elif m_name.find("RPN.FPN_RPN_conv") >= 0:
        s = "proposal_generator.rpn_head.conv." + parts[-1]
    elif m_name.find("RPN.FPN_RPN_cls_score") >= 0:
        s = "proposal_generator.rpn_head.objectness_logits." + parts[-1]
    elif m_name.find("RPN.FPN_RPN_bbox_pred") >= 0:
        s = "proposal_generator.rpn_head.anchor_deltas." + parts[-1]
    elif m_name.find("Box_Head") >= 0:
        s = "roi_heads.box_head." + parts[-2] + "." + parts[-1]
    elif m_name.find("Box_Outs") >= 0:
        s = "roi_heads.box_predictor." + parts[-2] + "." + parts[-1]
    return s


def make_backbone_map(m_name):
    return "backbone." + m_name


def checkpoint_key(k, ckpt):
    """The lookup order of load_ckpt_rel (:215-222): own name, res101 map, res50-FPN map, `backbone.` prefix."""
    for kk in (k, make_res101_map(k), make_res50_fpn_map(k), make_backbone_map(k)):
        if kk in ckpt:
            return kk
    return None


def trans_prd(ckpt, find_s="Prd_RCNN."):
    """net_rel.py:192-209: keep the entries under `find_s`, with the prefix (and anything before it) removed."""
    if not any(k.find(find_s) >= 0 for k in ckpt):
        return ckpt
    return {k[k.find(find_s) + len(find_s):]: np.asarray(v) for k, v in ckpt.items() if k.find(find_s) >= 0}


def load_ckpt_rel(model, ckpt):
    """net_rel.py:211-232: copy every state-dict entry whose (remapped) name is in `ckpt` (a dict of arrays / tensors)
    and whose shape matches (see the note on the class-count mismatch branch below).
    Returns the names that were loaded.  Folded / packed copies of the weights are dropped afterwards."""
    from .resnet3d import _PACK_EPOCH
    loaded = []
    with torch.no_grad():
        for k, v in model.state_dict().items():
            kk = checkpoint_key(k, ckpt)
            if kk is None:
                continue
            src = ckpt[kk]
            param = (src.detach().cpu() if torch.is_tensor(src) else torch.from_numpy(np.asarray(src))).to(v.device)
            if v.shape == param.shape:
                v.copy_(param)
                loaded.append(k)
            # A detector head with another class count (:226-227): the reference writes `v[c2a_idx].copy_(param[c2a])`,
            # and indexing a tensor with an index ARRAY returns a copy - the rows land in a temporary and `v` keeps its
            # initial values (measured: tests/golden/py_eval.npz tiny_sd_Box_Outs.*).  Same outcome here: not loaded.
    _PACK_EPOCH[0] += 1
    return loaded


# ------------------------------------------------------------------------------------------ rel_detections.pkl schema
def det_results(return_dict, im_scale=1.0, add_scores_all=False, use_freq_bias=False, use_spatial_feat=False):
    """test_rel.py:117-147: the network's return dict -> the per-frame record the evaluation reads.  Boxes go back to
    the original image (/ im_scale), labels become 0-based int32; `{}` when the frame produced no pair."""
    def arr(v):
        return v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
    if return_dict.get("sbj_rois") is None:
        return {}
    out = dict(sbj_boxes=arr(return_dict["sbj_rois"])[:, 1:5] / im_scale,
               sbj_labels=(arr(return_dict["sbj_labels"]) - 1).astype(np.int32, copy=False),
               sbj_scores=arr(return_dict["sbj_scores"]),
               obj_boxes=arr(return_dict["obj_rois"])[:, 1:5] / im_scale,
               obj_labels=(arr(return_dict["obj_labels"]) - 1).astype(np.int32, copy=False),
               obj_scores=arr(return_dict["obj_scores"]),
               prd_scores=arr(return_dict["prd_scores"]))
    if add_scores_all:
        out["prd_scores_ttl"] = arr(return_dict["prd_ttl_scores"])
    if use_freq_bias:
        out["prd_scores_bias"] = arr(return_dict["prd_scores_bias"])
    if use_spatial_feat:
        out["prd_scores_spt"] = arr(return_dict["prd_scores_spt"])
    return out


def frame_result(det, entry, do_val=True):
    """test_engine_rel.py:308-317: add the frame's name and, for validation, its ground truth (roidb entry keys)."""
    res = dict(det)
    res["image"] = entry["image"]
    if do_val:
        res.update(gt_sbj_boxes=entry["sbj_gt_boxes"], gt_sbj_labels=entry["sbj_gt_classes"],
                   gt_obj_boxes=entry["obj_gt_boxes"], gt_obj_labels=entry["obj_gt_classes"],
                   gt_prd_labels=entry["prd_gt_classes"])
    return res


def det_file_name(ind_range=None, use_gt_boxes=False, use_gt_labels=False):
    """test_engine_rel.py:340-351."""
    if ind_range is not None:
        return "rel_detection_range_%s_%s.pkl" % tuple(ind_range)
    if use_gt_boxes:
        return "rel_detections_gt_boxes_prdcls.pkl" if use_gt_labels else "rel_detections_gt_boxes_sgcls.pkl"
    return "rel_detections.pkl"


def save_detections(all_results, output_dir, ind_range=None, use_gt_boxes=False, use_gt_labels=False):
    """test_engine_rel.py:352 / utils/io.py:39-43: ONE pickle (highest protocol) of the list of per-frame dicts."""
    import os
    import pickle
    path = os.path.abspath(os.path.join(output_dir, det_file_name(ind_range, use_gt_boxes, use_gt_labels)))
    with open(path, "wb") as fh:
        pickle.dump(list(all_results), fh, pickle.HIGHEST_PROTOCOL)
    return path


def load_detections(path):
    """Read a rel_detections pickle: lists / dicts / strings / numbers / numpy arrays only - the unpickler resolves
    numpy's array reconstruction helpers and nothing else, so a file of unknown origin cannot run code."""
    import pickle

    allowed = {("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
               ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
               ("numpy.core.numeric", "_frombuffer"), ("numpy._core.numeric", "_frombuffer"),
               ("numpy", "ndarray"), ("numpy", "dtype")}

    class _ArraysOnly(pickle.Unpickler):
        def find_class(self, module, name):
            if (module, name) in allowed:
                return super().find_class(module, name)
            raise pickle.UnpicklingError("rel_detections: unexpected object %s.%s" % (module, name))

    with open(path, "rb") as fh:
        return _ArraysOnly(fh).load()
