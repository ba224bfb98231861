"""The inference loop around the hot path (trace_amd/test_engine.py = lib/core/test_engine_rel.py:215-357, :420-520 and
lib/core/test_rel.py:55-147, :275-302) on the CPU with a stub network: roidb -> frame window -> blobs -> model(**inputs)
-> per-frame record -> rel_detections pickle -> frame recall.  A stub that answers with the frame's own annotation must
come out with recall 1."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


class StubNet:
    """Checks what the loop hands to the network (the reference wraps every input in a one-element list for
    DataParallel's scatter, test_rel.py:82-113) and returns the annotated pairs as perfect detections."""

    def __init__(self, cfg):
        self.cfg, self.calls = cfg, []

    def __call__(self, data, im_info, dataset_name, do_vis, pre_processed_temporal_roi, pre_processed_frames_rpn_ret,
                 bf_cur_len, f_scale, all_frames, roidb=None, use_gt_labels=None):
        for v in (data, im_info, dataset_name, do_vis, bf_cur_len, f_scale, all_frames, roidb):
            assert isinstance(v, list) and len(v) == 1
        data, im_info, all_frames, roidb = data[0], im_info[0], all_frames[0], roidb[0]
        assert data.dtype == torch.float32 and data.shape[:2] == (1, 3)
        assert tuple(im_info.shape) == (1, 3) and float(im_info[0, 0]) == data.shape[2] and float(im_info[0, 1]) == data.shape[3]
        T = all_frames.shape[1]
        assert all_frames.dim() == 5 and all_frames.shape[0] == 1 and all_frames.shape[2] == 3
        assert 1 <= T <= 2 * self.cfg.HALF_NUMBER_OF_FRAMES + 1 and 0 <= bf_cur_len[0] < T
        self.calls.append((T, bf_cur_len[0], f_scale[0], float(im_info[0, 2])))
        scale = float(im_info[0, 2])
        P = roidb["sbj_gt_boxes"].shape[0]
        prd = np.full((P, self.cfg.NUM_PRD_CLASSES), 0.01, np.float32)
        prd[np.arange(P), roidb["prd_gt_classes"]] = 0.9
        z = np.zeros((P, 1), np.float32)
        return dict(sbj_rois=np.hstack((z, roidb["sbj_gt_boxes"] * scale)), obj_rois=np.hstack((z, roidb["obj_gt_boxes"] * scale)),
                    sbj_labels=torch.from_numpy(roidb["sbj_gt_classes"] + 1), obj_labels=torch.from_numpy(roidb["obj_gt_classes"] + 1),
                    sbj_scores=torch.ones(P), obj_scores=torch.ones(P), prd_scores=torch.from_numpy(prd),
                    prd_ttl_scores=torch.from_numpy(prd), prd_scores_bias=torch.from_numpy(prd))


def test_test_net_loop_end_to_end(tmp_path):
    from test_net import synthetic_roidb
    from trace_amd import evaluation, test_engine
    from trace_amd.config import vidvrd_config
    from trace_amd.io import load_detections
    cfg = vidvrd_config()
    cfg.TEST_SCALE, cfg.TEST_MAX_SIZE = 320, 1000                # key frame resized 224 -> 320, clip frames stay 224
    roidb, read_frame = synthetic_roidb(2, 3, 8, 6, hw=224)
    net = StubNet(cfg)
    results, path = test_engine.test_net(net, roidb, "vidvrd_val", str(tmp_path), read_frame, cfg, do_val=True,
                                         use_gt_boxes=True, use_gt_labels=False)
    assert os.path.basename(path) == "rel_detections_gt_boxes_sgcls.pkl" and len(results) == 6
    # key frames 20, 35, 50 at interval 4 with frame 0 as the first: 5 frames before the first key frame, then 8 / 8
    assert [c[:2] for c in net.calls[:3]] == [(14, 5), (17, 8), (17, 8)]
    assert all(abs(c[2] - 1.0) < 1e-12 and abs(c[3] - 320.0 / 224.0) < 1e-6 for c in net.calls)
    back = load_detections(path)
    assert [r["image"] for r in back] == [e["image"] for e in roidb]
    for r, e in zip(back, roidb):
        np.testing.assert_allclose(r["sbj_boxes"], e["sbj_gt_boxes"], atol=1e-3)      # boxes back in image scale
        assert np.array_equal(r["sbj_labels"], e["sbj_gt_classes"]) and r["sbj_labels"].dtype == np.int32
        assert np.array_equal(r["gt_prd_labels"], e["prd_gt_classes"])
    image_recall, video_recall, _ = evaluation.eval_frame_recall(back, topk=100, prd_k=20)
    assert image_recall[100] > 0.999 and video_recall[100] > 0.999


def test_frames_blob_stops_at_a_missing_neighbour():
    """test_engine_rel.py:465-484: the clip ends at the first neighbour frame that does not exist."""
    from trace_amd import test_engine
    from trace_amd.config import vidvrd_config
    cfg = vidvrd_config()
    frame = np.full((224, 224, 3), 7, np.uint8)
    have = {"/d/sampled_frames/v/%06d.jpg" % j for j in (0, 4, 8, 12, 16, 24, 28)}       # 32 is missing

    def read_frame(p):
        return frame if (p in have or p == "/d/frames/v/000020.jpg") else None
    blob, bf, fs = test_engine.get_frames_blob("/d/frames/v/000020.jpg", frame, cfg, read_frame, im_id_st=0)
    assert tuple(blob.shape) == (1, 8, 3, 224, 224) and bf == 5 and fs == 1.0
    np.testing.assert_allclose(blob[0, :, 0].numpy(), 7 - 102.9801, rtol=1e-6)
