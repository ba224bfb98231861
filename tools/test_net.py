#!/usr/bin/env python3
"""test_net_rel.py's inner loop end to end on SYNTHETIC frames (no dataset offline): a roidb of key frames with
annotated (subject, object, predicate) boxes -> trace_amd.test_engine.test_net (frame window, blobs, model, record,
rel_detections pickle) -> frame recall (trace_amd.evaluation.eval_frame_recall).  Reference: tools/test_net_rel.py:90-334,
lib/core/test_engine_rel.py:85-357, lib/datasets_rel/task_evaluation_vg_and_vrd.py:52-330.

    python tools/test_net.py --videos 2 --frames-per-video 3 --out /tmp/trace_out         (one MI355X)
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def synthetic_roidb(n_videos, keys_per_video, n_boxes, n_rel, hw=224, seed=1, n_cls=35, n_prd=132, root="/data/vidvrd"):
    """Key-frame entries (roidb keys of lib/datasets_rel/json_dataset_rel.py) + an in-memory frame store."""
    from trace_amd.synthetic import synth_boxes
    rng = np.random.default_rng(seed)
    roidb, store = [], {}
    for v in range(n_videos):
        vid = "video_%03d" % v
        n_frames = 15 * keys_per_video + 40
        for f in range(n_frames):
            store["%s/sampled_frames/%s/%06d.jpg" % (root, vid, f)] = (seed, v, f)
        for k in range(keys_per_video):
            fid = 15 * k + 20
            path = "%s/frames/%s/%06d.jpg" % (root, vid, fid)
            store[path] = (seed, v, fid)
            boxes = np.round(synth_boxes(rng, n_boxes, 2.0, float(hw)))
            cls = rng.integers(0, n_cls, n_boxes)
            s = rng.integers(0, n_boxes, n_rel)
            o = (s + rng.integers(1, n_boxes, n_rel)) % n_boxes
            roidb.append(dict(image=path, sbj_gt_boxes=boxes[s].astype(np.float32), obj_gt_boxes=boxes[o].astype(np.float32),
                              sbj_gt_classes=cls[s].astype(np.int32), obj_gt_classes=cls[o].astype(np.int32),
                              prd_gt_classes=rng.integers(0, n_prd, n_rel).astype(np.int32)))

    def read_frame(path, hw=hw):
        key = store.get(path)
        if key is None:
            return None
        g = np.random.default_rng(hash(key) & 0x7fffffff)
        return g.integers(0, 256, (hw, hw, 3), dtype=np.uint8)
    return roidb, read_frame


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--videos", type=int, default=2)
    ap.add_argument("--frames-per-video", type=int, default=3)
    ap.add_argument("--boxes", type=int, default=8)
    ap.add_argument("--relations", type=int, default=6)
    ap.add_argument("--out", default="/tmp/trace_test_net")
    ap.add_argument("--use-gt-labels", action="store_true")
    args = ap.parse_args()
    from trace_amd import evaluation, test_engine
    from trace_amd.io import load_detections
    from trace_amd.model_builder_rel import Generalized_RCNN
    from trace_amd.resnet2d import ResNet50_conv5_body_dilation2, ResNet101_conv5_body
    from trace_amd.synthetic import fill_resnet2d_state, randomize_model
    dev = torch.device("cuda", 0)
    torch.manual_seed(1)
    m = Generalized_RCNN()
    randomize_model(m, 1)
    fill_resnet2d_state(m.Conv_Body, 1)
    fill_resnet2d_state(m.Prd_RCNN2D, 2)
    m.eval().to(dev)
    m.Prd_RCNN.calibrate_bn((torch.randn((1, 3, 16, 112, 112), generator=torch.Generator().manual_seed(1)) * 50).to(dev))
    cfg = m.cfg
    cfg.TEST_SCALE, cfg.TEST_MAX_SIZE = 224, 1000          # synthetic frames are 224^2: no upscaling of the key frame
    roidb, read_frame = synthetic_roidb(args.videos, args.frames_per_video, args.boxes, args.relations)
    results, path = test_engine.test_net(m, roidb, "vidvrd_val", args.out, read_frame, cfg, do_val=True, use_gt_boxes=True,
                                         use_gt_labels=args.use_gt_labels, device=dev)
    back = load_detections(path)
    assert len(back) == len(roidb) and all(np.array_equal(a["prd_scores"], b["prd_scores"]) for a, b in zip(back, results))
    image_recall, video_recall, _ = evaluation.eval_frame_recall([r for r in back if r is not None], topk=100, prd_k=20)
    print(json.dumps(dict(frames=len(back), file=path, pairs=[int(r["prd_scores"].shape[0]) for r in back],
                          image_recall=image_recall, video_recall=video_recall)))


if __name__ == "__main__":
    main()
