"""The reference's single-GPU inference loop around the hot path:

  test_net          lib/core/test_engine_rel.py:215-357   roidb -> per key frame: frame window -> blobs -> model -> record
                                                         -> one rel_detections*.pkl
  get_frames_blob   lib/core/test_engine_rel.py:420-520   the clip around the key frame (io.frame_window / clip_from_frames)
  im_get_det_rels   lib/core/test_rel.py:60-147           key-frame blob + im_info, model(**inputs), record (io.det_results)
  _get_blobs        lib/core/test_rel.py:275-306          (non-FPN branch)

Decoding is the caller's: `read_frame(path)` returns a BGR uint8 array (H,W,3) or None when the file does not exist -
cv2.imread and os.path.exists of the reference (OpenCV is not in this image).  The model is anything with the
reference's forward signature: trace_amd.Generalized_RCNN on the GPU, or a stub in the CPU test of this loop."""
import os

import numpy as np
import torch

from . import io as tio

SEG_STEP = 15                    # test_engine_rel.py:237-239
N_LIST = tuple(range(30))


def is_vidvrd_key_frame(img_name):
    """test_engine_rel.py:246-253 (with N_list = 0..29 every frame id passes; kept as the reference computes it)."""
    cur = int(img_name.split('/')[-1].split('.')[0])
    return any((cur - n) % SEG_STEP == 0 and cur >= n for n in N_LIST)


def neighbour_frame_path(cur_image_abs_path, j):
    """test_engine_rel.py:421-430, :459: <root>/sampled_frames/<video>/<%06d>.jpg, root = three levels above the frame."""
    parts = cur_image_abs_path.split('/')
    video_path = '/'
    for p in parts[:-3]:
        video_path = os.path.join(video_path, p)
    return os.path.join(video_path, 'sampled_frames', parts[-2], '{:06d}'.format(j) + '.jpg')


def get_frames_blob(cur_image_abs_path, im0, cfg, read_frame, im_id_st=1, device=None):
    """test_engine_rel.py:420-520 -> (all_frames (1,T,3,h,w) f32, bf_cur_len, f_scale).  The key frame itself is read
    again from its own path, as the reference does (:463); the walk stops at the first missing neighbour (:482-484)."""
    frame_id = int(cur_image_abs_path.split('/')[-1].split('.')[0])
    cache = {}

    def path_of(j):
        return cur_image_abs_path if j == frame_id else neighbour_frame_path(cur_image_abs_path, j)

    def exists(j):
        if j not in cache:
            cache[j] = im0 if (j == frame_id and im0 is not None) else read_frame(path_of(j))
        return cache[j] is not None
    ids, bf_cur_len = tio.frame_window(frame_id, cfg.HALF_NUMBER_OF_FRAMES, cfg.FRAMES_INTERVAL, im_id_st, exists)
    frames = [cache[j] for j in ids]
    blob, f_scale = tio.clip_from_frames(frames, cfg.TEMPORAL_SCALES, 1000, device=device)
    return blob, bf_cur_len, f_scale


def get_image_blob(im, target_scale, target_max_size, device=None):
    """test_rel.py:275-302, non-FPN branch: (data (1,3,H,W) f32 mean-subtracted + resized, im_info (1,3), im_scale)."""
    data, im_scale = tio.clip_from_frames(im[None], target_scale, target_max_size, device=device)
    data = data[0]
    im_info = np.hstack((data.shape[2], data.shape[3], im_scale))[np.newaxis, :].astype(np.float32)
    return data, torch.from_numpy(im_info), im_scale


def im_detect_rels(model, im, all_frames, dataset_name, cfg, roidb=None, use_gt_labels=False, frames_rpn_ret=None,
                   temporal_roi=None, bf_cur_len=None, f_scale=None, device=None):
    """test_rel.py:55-147: network inputs (each wrapped in a one-element list, :82-113), forward, per-frame record."""
    data, im_info, im_scale = get_image_blob(im, cfg.TEST_SCALE, cfg.TEST_MAX_SIZE, device)
    inputs = dict(data=[data], im_info=[im_info], dataset_name=[dataset_name], do_vis=[False],
                  pre_processed_temporal_roi=[temporal_roi], pre_processed_frames_rpn_ret=[frames_rpn_ret],
                  bf_cur_len=[bf_cur_len], f_scale=[f_scale], all_frames=[all_frames])
    if roidb is not None:
        inputs['roidb'] = [roidb]
    if use_gt_labels:
        inputs['use_gt_labels'] = [use_gt_labels]
    return_dict = model(**inputs)
    return tio.det_results(return_dict, im_scale, add_scores_all=True, use_freq_bias=getattr(cfg, "USE_FREQ_BIAS", False))


def test_net(model, roidb, dataset_name, output_dir, read_frame, cfg, do_val=True, use_gt_boxes=False,
             use_gt_labels=False, ind_range=None, device=None, log=None):
    """test_engine_rel.py:215-357 over `roidb` (already cut to ind_range by the caller, as get_roidb_and_dataset does).
    Writes ONE pickle named as the reference names it and returns the list of per-frame records (None for skipped
    frames, as the reference leaves them)."""
    all_results = [None for _ in range(len(roidb))]
    im_id_st = 0 if dataset_name.find('vidvrd') >= 0 else 1                      # :275-286
    for i, entry in enumerate(roidb):
        img_name = entry['image']
        if dataset_name.find('vidvrd') >= 0 and not is_vidvrd_key_frame(img_name):
            continue
        frames_rpn_ret = entry.get('pre_processed_frames_rpn_ret')             # ENABLE_FRAME_PRE_PROCESSING (:255-261)
        temporal_roi = entry.get('pre_processed_temporal_roi')
        im = read_frame(img_name)
        all_frames, bf_cur_len, f_scale = get_frames_blob(img_name, im, cfg, read_frame, im_id_st, device)
        res = im_detect_rels(model, im, all_frames, dataset_name, cfg, entry if use_gt_boxes else None,
                             use_gt_labels if use_gt_boxes else False, frames_rpn_ret, temporal_roi, bf_cur_len, f_scale, device)
        all_results[i] = tio.frame_result(res, entry, do_val)
        if log is not None:
            log(i, len(roidb), img_name)
    os.makedirs(output_dir, exist_ok=True)
    path = tio.save_detections(all_results, output_dir, ind_range, use_gt_boxes, use_gt_labels)
    return all_results, path
