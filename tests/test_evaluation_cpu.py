"""trace_amd.evaluation against the outputs of the reference's own evaluation code (tests/golden/py_eval.npz,
tools/make_goldens_r2c.py); the video association (dlib-dependent in the reference: parity unpinned) through its
defining properties."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = np.load(os.path.join(ROOT, "tests", "golden", "py_eval.npz"))


def jl(a):
    return json.loads(bytes(a).decode())


def test_frame_topk_and_matches_equal_reference():
    from trace_amd import evaluation as E
    for ci, case in enumerate(jl(G["frame_cases"])):
        key = "frame%d_" % ci
        res = {k[len(key) + 3:]: G[k] for k in G.files if k.startswith(key + "in_")}
        top = E.topk_frame_triplets(res, case["prd_k"], case["topk"])
        for k in ("det_boxes_s_top", "det_boxes_o_top", "det_labels_s_top", "det_labels_p_top", "det_labels_o_top", "det_scores_top"):
            assert np.array_equal(top[k], G[key + k]), (ci, k)
        gt_so = np.hstack((res["gt_sbj_boxes"], res["gt_obj_boxes"]))
        gt_spo = np.vstack((res["gt_sbj_labels"], res["gt_prd_labels"], res["gt_obj_labels"])).transpose()
        det_spo = np.vstack((top["det_labels_s_top"], top["det_labels_p_top"], top["det_labels_o_top"])).transpose()
        p2g = E.compute_pred_matches(gt_spo, det_spo, gt_so, np.hstack((top["det_boxes_s_top"], top["det_boxes_o_top"])), 0.5, False)
        assert p2g == jl(G[key + "p2g"])
        p2g_phr = E.compute_pred_matches(gt_spo, det_spo, E.boxes_union(res["gt_sbj_boxes"], res["gt_obj_boxes"]),
                                         E.boxes_union(top["det_boxes_s_top"], top["det_boxes_o_top"]), 0.5, True)
        assert p2g_phr == jl(G[key + "p2g_phr"])
        # recall over this one frame = matched gt relations / gt relations
        res["image"] = "data/vidvrd/frames/v%d/000015.jpg" % ci
        img_r, vid_r, dets = E.eval_frame_recall([res], topk=case["topk"], prd_k=case["prd_k"], recall_ks=(20, 50, 100))
        for k in (20, 50, 100):
            want = len(set(g for m in p2g[:k] for g in m)) / (gt_spo.shape[0] + 1e-12)
            assert abs(img_r[k] - want) < 1e-12 and abs(vid_r[k] - want) < 1e-9


def test_bbox_overlaps_matches_compiled_reference_kernel():
    from oracle import c_oracle
    from trace_amd.evaluation import bbox_overlaps
    rng = np.random.default_rng(2)
    a = np.hstack((rng.uniform(0, 100, (50, 2)), rng.uniform(100, 220, (50, 2)))).astype(np.float32)
    b = np.hstack((rng.uniform(0, 150, (70, 2)), rng.uniform(150, 300, (70, 2)))).astype(np.float32)
    b[:5] = a[:5]
    np.testing.assert_allclose(bbox_overlaps(a, b), c_oracle.bbox_overlaps(a, b), rtol=0, atol=1e-6)


def test_vidvrd_metrics_equal_reference():
    from trace_amd import evaluation as E
    gt, pred, want = jl(G["vid_gt"]), jl(G["vid_pred"]), jl(G["vid_metrics"])
    mean_ap, rec, prec = E.evaluate(gt, pred)
    assert abs(mean_ap - want["mean_ap"]) < 1e-12
    assert {str(k): float(v) for k, v in rec.items()} == want["rec"]
    assert {str(k): float(v) for k, v in prec.items()} == want["prec"]
    p, r, h = E.eval_detection_scores(gt["vid01"], pred["vid01"], 0.5)
    assert np.array_equal(p, G["vid01_prec"]) and np.array_equal(r, G["vid01_rec"]) and np.array_equal(h, G["vid01_hit"])


def test_association_merges_overlapping_segments():
    """Parity unpinned (the reference's association needs dlib): checked through what the algorithm promises -
    the same triplet on overlapping trajectories in consecutive segments becomes ONE video relation spanning both, with
    the best score and averaged boxes on the shared frames; a different triplet or a far-away trajectory starts a new one."""
    from trace_amd.evaluation import Trajectory, greedy_relational_association

    def box_track(n, x0):
        return [(x0 + i, 10.0, x0 + 50 + i, 90.0) for i in range(n)]

    tracks = {0: [Trajectory(0, 30, box_track(30, 0)), Trajectory(0, 30, box_track(30, 200))],
              15: [Trajectory(15, 45, box_track(30, 15)), Trajectory(15, 45, box_track(30, 215)), Trajectory(15, 45, box_track(30, 400))]}
    rel = [(("v", 0, 30), [(0.6, (1, 2, 3), (0, 1)), (0.3, (1, 0, 3), (0, 1))]),
           (("v", 15, 45), [(0.9, (1, 2, 3), (0, 1)), (0.5, (1, 2, 3), (2, 1)), (0.2, (3, 2, 1), (1, 0))])]
    out = greedy_relational_association({1: "a", 3: "b"}, {0: "p0", 2: "p2"}, rel, lambda vid, s, e: tracks[s])
    merged = [r for r in out if r["triplet"] == ["a", "p2", "b"] and r["duration"] == [0, 45]]
    assert len(merged) == 1 and merged[0]["score"] == 0.9 and len(merged[0]["sub_traj"]) == 45
    assert merged[0]["sub_traj"][20] == (20.0, 10.0, 70.0, 90.0)          # both segments agree on the shared frames
    assert sorted((r["triplet"], tuple(r["duration"])) for r in out if r is not merged[0]) == sorted([
        (["a", "p0", "b"], (0, 30)), (["a", "p2", "b"], (15, 45)), (["b", "p2", "a"], (15, 45))])
