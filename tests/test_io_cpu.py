"""trace_amd.io: checkpoint key maps / load_ckpt_rel against the reference's own functions
(tests/golden/py_eval.npz ckpt_* / tiny_*, tools/make_goldens_r2c.py); frame window and blob preparation (the
reference's path needs OpenCV: parity unpinned, checked through its defining properties)."""
import json
import os

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = np.load(os.path.join(ROOT, "tests", "golden", "py_eval.npz"))


def jl(a):
    return json.loads(bytes(a).decode())


def test_key_maps_equal_reference_on_every_model_key():
    from trace_amd import io
    keys = jl(G["ckpt_keys"])
    assert [io.make_res101_map(k) for k in keys] == jl(G["ckpt_res101"])
    assert [io.make_res50_fpn_map(k) for k in keys] == jl(G["ckpt_res50fpn"])
    assert [io.make_backbone_map(k) for k in keys] == jl(G["ckpt_backbone"])
    c2a, idx = io.class_coco2ag()
    assert np.array_equal(c2a, G["c2a"]) and np.array_equal(idx, G["c2a_idx"])
    # the keys the generator used ARE this model's keys (plus the FPN-style extras at the end)
    from trace_amd.model_builder_rel import Generalized_RCNN
    from trace_amd.rpn import single_scale_rpn_outputs
    m = Generalized_RCNN()
    m.RPN = single_scale_rpn_outputs(2048, 1 / 16)
    mine = list(m.state_dict().keys())
    assert keys[:len(mine)] == mine


def test_load_ckpt_rel_equals_reference():
    from trace_amd import io

    class Tiny(nn.Module):
        def __init__(self):
            super().__init__()
            self.Box_Outs = nn.Module()
            self.Box_Outs.cls_score = nn.Linear(8, 37)
            self.Box_Head = nn.Module()
            self.Box_Head.fc1 = nn.Linear(8, 4)
            self.other = nn.Linear(3, 3)
    t = Tiny()
    for p in t.parameters():
        p.data.zero_()
    ck = {k[len("tiny_ck_"):]: G[k] for k in G.files if k.startswith("tiny_ck_")}
    loaded = io.load_ckpt_rel(t, ck)
    for k, v in t.state_dict().items():
        assert np.array_equal(v.numpy(), G["tiny_sd_" + k]), k
    assert "Box_Head.fc1.bias" not in loaded and "other.weight" in loaded and "Box_Head.fc1.weight" in loaded
    assert "Box_Outs.cls_score.weight" not in loaded        # 81 COCO rows vs 37: the reference's re-indexing writes to a copy
    assert io.trans_prd({"module.Prd_RCNN.conv1.conv.weight": np.zeros(2), "x": np.ones(1)}).keys() == {"conv1.conv.weight"}


def test_frame_window():
    from trace_amd.io import frame_window
    # VidVRD config: HALF_NUMBER_OF_FRAMES 8, FRAMES_INTERVAL 4 -> 17 frames, key frame in the middle
    ids, bf = frame_window(100, 8, 4)
    assert ids == list(range(68, 133, 4)) and bf == 8 and ids[bf] == 100
    # near the start of the video: fewer frames before the key frame
    ids, bf = frame_window(9, 8, 4, im_id_st=1)
    assert ids == [1, 5, 9] + list(range(13, 42, 4)) and bf == 2 and ids[bf] == 9
    # frames missing after the key frame: the walk stops at the first gap
    ids, bf = frame_window(100, 8, 4, exists=lambda j: j <= 112)
    assert ids == list(range(68, 113, 4)) and bf == 8
    # the reference's defaults (3, 3)
    ids, bf = frame_window(30, 3, 3)
    assert ids == [21, 24, 27, 30, 33, 36, 39] and bf == 3


def test_clip_from_frames():
    from trace_amd.io import PIXEL_MEANS, clip_from_frames, get_target_scale
    assert get_target_scale(240, 320, 320, 1000) == 320 / 240 and get_target_scale(500, 2000, 800, 1333) == 1333 / 2000
    rng = np.random.default_rng(0)
    fr = rng.integers(0, 256, (5, 224, 224, 3), dtype=np.uint8)
    blob, s = clip_from_frames(fr, 224)
    assert s == 1.0 and blob.shape == (1, 5, 3, 224, 224) and blob.dtype == torch.float32
    want = fr.astype(np.float32) - np.asarray(PIXEL_MEANS, np.float32)
    assert np.array_equal(blob[0].permute(0, 2, 3, 1).numpy(), want)
    u8, s = clip_from_frames(list(fr), 224, keep_uint8=True)
    assert u8.dtype == torch.uint8 and u8.shape == (1, 5, 224, 224, 3) and np.array_equal(u8[0].numpy(), fr)
    # resize: output size = round(size * scale); a constant image stays constant; means are subtracted before resizing
    fr2 = np.full((2, 240, 320, 3), 200, np.uint8)
    blob, s = clip_from_frames(fr2, 320)
    assert s == 320 / 240 and blob.shape == (1, 2, 3, 320, 427)
    np.testing.assert_allclose(blob[0, :, 1].numpy(), 200 - PIXEL_MEANS[1], atol=1e-4)


def test_rel_detections_schema_round_trip_and_recall(tmp_path):
    """test_rel.py:117-147 + test_engine_rel.py:308-352 (restated; the reference's function needs its cfg / OpenCV blob
    path around it: parity unpinned): record keys, dtypes, label / box conventions, file names, and that the loaded
    pickle is what `evaluation.eval_frame_recall` consumes."""
    import pickle
    from trace_amd import io, evaluation
    rng = np.random.RandomState(0)
    results = []
    for i in range(3):
        P = 6
        sb = np.sort(rng.rand(P, 4).astype(np.float32) * 200, axis=1)
        ob = np.sort(rng.rand(P, 4).astype(np.float32) * 200, axis=1)
        rd = dict(sbj_rois=torch.from_numpy(np.hstack([np.zeros((P, 1), np.float32), sb * 1.5])),
                  obj_rois=torch.from_numpy(np.hstack([np.zeros((P, 1), np.float32), ob * 1.5])),
                  sbj_labels=torch.from_numpy(rng.randint(1, 36, P).astype(np.int64)),
                  obj_labels=torch.from_numpy(rng.randint(1, 36, P).astype(np.int64)),
                  sbj_scores=torch.from_numpy(rng.rand(P).astype(np.float32)),
                  obj_scores=torch.from_numpy(rng.rand(P).astype(np.float32)),
                  prd_scores=torch.from_numpy(rng.rand(P, 133).astype(np.float32)),
                  prd_ttl_scores=torch.from_numpy(rng.rand(P, 133).astype(np.float32)))
        det = io.det_results(rd, im_scale=1.5, add_scores_all=True)
        assert set(det) == {"sbj_boxes", "sbj_labels", "sbj_scores", "obj_boxes", "obj_labels", "obj_scores", "prd_scores",
                            "prd_scores_ttl"}
        assert det["sbj_labels"].dtype == np.int32 and det["obj_labels"].dtype == np.int32
        assert np.array_equal(det["sbj_labels"], rd["sbj_labels"].numpy() - 1)
        np.testing.assert_allclose(det["sbj_boxes"], sb, rtol=1e-6)
        np.testing.assert_allclose(det["obj_boxes"], ob, rtol=1e-6)
        # ground truth = the first two predicted pairs, so recall must be 1 at any k
        entry = dict(image="vid%d/%06d.jpg" % (i // 2, i), sbj_gt_boxes=det["sbj_boxes"][:2].copy(), obj_gt_boxes=det["obj_boxes"][:2].copy(),
                     sbj_gt_classes=det["sbj_labels"][:2].copy(), obj_gt_classes=det["obj_labels"][:2].copy(),
                     prd_gt_classes=np.argmax(det["prd_scores"][:2, 1:], axis=1).astype(np.int32))
        res = io.frame_result(det, entry, do_val=True)
        assert res["image"] == entry["image"] and set(res) - set(det) == {"image", "gt_sbj_boxes", "gt_sbj_labels", "gt_obj_boxes",
                                                                          "gt_obj_labels", "gt_prd_labels"}
        results.append(res)
    assert io.det_results(dict(sbj_rois=None)) == {}
    assert "gt_sbj_boxes" not in io.frame_result(det, entry, do_val=False)
    assert io.det_file_name() == "rel_detections.pkl"
    assert io.det_file_name(use_gt_boxes=True) == "rel_detections_gt_boxes_sgcls.pkl"
    assert io.det_file_name(use_gt_boxes=True, use_gt_labels=True) == "rel_detections_gt_boxes_prdcls.pkl"
    assert io.det_file_name(ind_range=(10, 20)) == "rel_detection_range_10_20.pkl"
    path = io.save_detections(results, str(tmp_path), use_gt_boxes=True)
    assert path.endswith("rel_detections_gt_boxes_sgcls.pkl")
    back = io.load_detections(path)
    assert len(back) == 3 and all(np.array_equal(a[k], b[k]) if isinstance(a[k], np.ndarray) else a[k] == b[k]
                                  for a, b in zip(results, back) for k in a)
    # a pickle that names anything but numpy's array helpers is refused
    evil = tmp_path / "evil.pkl"
    evil.write_bytes(pickle.dumps([os.path.join]))
    try:
        io.load_detections(str(evil))
        raise AssertionError("load_detections accepted a pickle with a callable in it")
    except pickle.UnpicklingError:
        pass
    # the loaded list is the evaluation's input: highest-scoring predicate of each gt pair is its gt label -> found
    img_rec, vid_rec, top = evaluation.eval_frame_recall(back, topk=100, prd_k=1, recall_ks=(20, 50))
    assert len(top) == 3 and 0.0 <= img_rec[20] <= 1.0 and img_rec[50] >= img_rec[20]
