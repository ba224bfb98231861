"""Fixtures for the evaluation code, produced by RUNNING THE REFERENCE'S OWN PYTHON in the build container
(environment: tools/ref_env.py); data only.

    python tools/make_goldens_r2c.py

  py_eval.npz
    frame_*   lib/datasets_rel/task_evaluation_vg_and_vrd.py: the per-frame top-k block (:168-249, inline in
              eval_rel_results: the generator reads exactly those lines from the reference file AT RUN TIME, dedents and
              executes them on a synthetic result dict) and `_compute_pred_matches` (:366-406) called directly, sgdet
              and phrdet.  `utils.boxes.bbox_overlaps` is Cython (not buildable here): the generator hands the
              reference module a numpy restatement of cython_bbox.pyx:32-73 (trace_amd.evaluation.bbox_overlaps), so
              these vectors pin everything around the IoU, not the IoU itself.
    vid_*     tools/transform_vidvrd_results.py `evaluate` (:209-265, with eval_detection_scores / eval_tagging_scores /
              viou / voc_ap) on synthetic ground-truth and predicted video relations.  The module imports dlib through
              vidvrd_utils.association at the top; dlib is absent, so an EMPTY placeholder module named dlib is
              registered (name only; nothing `evaluate` calls touches it).  The association itself
              (vidvrd_utils/association.py) needs dlib.drectangle at run time and stays unpinned.
    ckpt_*    lib/utils_rel/net_rel.py `make_res101_map` / `make_res50_fpn_map` / `make_backbone_map` / `class_coco2ag`
              on the model's state-dict keys, and `load_ckpt_rel` (:211-232) on a small module (tiny_*).
"""
import json
import os
import sys
import textwrap
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_env  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def jb(o):
    return np.frombuffer(json.dumps(o).encode(), dtype=np.uint8)


def synth_frame(rng, n_pairs, n_prd, n_gt):
    def boxes(n):
        xy = rng.uniform(0, 200, (n, 2))
        wh = rng.uniform(20, 120, (n, 2))
        return np.hstack((xy, xy + wh)).astype(np.float32)
    res = dict(sbj_boxes=boxes(n_pairs), obj_boxes=boxes(n_pairs), sbj_labels=rng.integers(0, 6, n_pairs).astype(np.int32),
               obj_labels=rng.integers(0, 6, n_pairs).astype(np.int32), sbj_scores=rng.uniform(0.1, 1, n_pairs).astype(np.float32),
               obj_scores=rng.uniform(0.1, 1, n_pairs).astype(np.float32), prd_scores=rng.uniform(0, 1, (n_pairs, n_prd)).astype(np.float32))
    # ground truth: jittered copies of some predicted pairs (so that matches exist) plus unrelated ones
    pick = rng.choice(n_pairs, n_gt, replace=False)
    res["gt_sbj_boxes"] = (res["sbj_boxes"][pick] + rng.uniform(-6, 6, (n_gt, 4))).astype(np.float32)
    res["gt_obj_boxes"] = (res["obj_boxes"][pick] + rng.uniform(-6, 6, (n_gt, 4))).astype(np.float32)
    res["gt_sbj_labels"] = res["sbj_labels"][pick].copy()
    res["gt_obj_labels"] = res["obj_labels"][pick].copy()
    res["gt_prd_labels"] = np.argmax(res["prd_scores"][pick], 1).astype(np.int32)
    res["gt_prd_labels"][::3] = (res["gt_prd_labels"][::3] + 1) % n_prd          # some wrong predicates
    return res


def frames(cfg, out):
    import datasets_rel.task_evaluation_vg_and_vrd as T
    from trace_amd.evaluation import bbox_overlaps as np_overlaps
    T.bbox_overlaps = np_overlaps
    src = open(os.path.join(ref_env.REF, "lib/datasets_rel/task_evaluation_vg_and_vrd.py")).read().split("\n")
    block = textwrap.dedent("\n".join(src[167:249]))       # lines 168..249
    cfg.MODEL.MULTI_RELATION = True
    rng = np.random.default_rng(5)
    cases = []
    for ci, (n_pairs, n_prd, n_gt, prd_k, topk) in enumerate(((40, 7, 9, 1, 100), (60, 5, 12, 2, 50), (12, 4, 3, 3, 20))):
        res = synth_frame(rng, n_pairs, n_prd, n_gt)
        ns = dict(res=res, np=np, cfg=cfg, prd_k=prd_k, topk=topk, argsort_desc=T.argsort_desc, cnt=0., alcnt=0.)
        exec(block, ns)
        key = "frame%d_" % ci
        for k, v in res.items():
            out[key + "in_" + k] = v
        for k in ("det_boxes_s_top", "det_boxes_o_top", "det_labels_s_top", "det_labels_p_top", "det_labels_o_top", "det_scores_top"):
            out[key + k] = np.asarray(ns[k])
        gt_so = np.hstack((res["gt_sbj_boxes"], res["gt_obj_boxes"]))
        gt_spo = np.vstack((res["gt_sbj_labels"], res["gt_prd_labels"], res["gt_obj_labels"])).transpose()
        for phr in (False, True):
            if phr:
                p2g = T._compute_pred_matches(gt_spo, ns["det_labels_spo_top"], T.boxes_union(res["gt_sbj_boxes"], res["gt_obj_boxes"]),
                                              T.boxes_union(ns["det_boxes_s_top"], ns["det_boxes_o_top"]), iou_thresh=0.5, phrdet=True)
            else:
                p2g = T._compute_pred_matches(gt_spo, ns["det_labels_spo_top"], gt_so, ns["det_boxes_so_top"], iou_thresh=0.5, phrdet=False)
            out[key + ("p2g_phr" if phr else "p2g")] = jb(p2g)
        cases.append(dict(prd_k=prd_k, topk=topk))
        print("frame case %d: %d top triplets, %d matched predictions" % (ci, len(ns["det_scores_top"]), sum(1 for m in p2g if m)))
    out["frame_cases"] = jb(cases)


def videos(out):
    sys.modules.setdefault("dlib", types.ModuleType("dlib"))
    sys.modules["dlib"].drectangle = None
    sys.modules["dlib"].correlation_tracker = None
    import transform_vidvrd_results as V
    rng = np.random.default_rng(9)

    def traj(n, x0, y0, jitter=0.0):
        return [[float(x0 + 2 * i + rng.uniform(-jitter, jitter)), float(y0 + i + rng.uniform(-jitter, jitter)),
                 float(x0 + 60 + 2 * i + rng.uniform(-jitter, jitter)), float(y0 + 80 + i + rng.uniform(-jitter, jitter))] for i in range(n)]
    names = ["person", "dog", "ball", "car"]
    preds_n = ["chase", "ride", "next_to"]
    gt, pr = {}, {}
    for v in range(4):
        vid = "vid%02d" % v
        gts, prs = [], []
        for r in range(5 + v):
            f0 = int(rng.integers(0, 40))
            ln = int(rng.integers(15, 45))
            trip = [names[int(rng.integers(0, 4))], preds_n[int(rng.integers(0, 3))], names[int(rng.integers(0, 4))]]
            x0, y0 = float(rng.uniform(0, 200)), float(rng.uniform(0, 100))
            g = dict(triplet=trip, duration=[f0, f0 + ln], sub_traj=traj(ln, x0, y0), obj_traj=traj(ln, x0 + 90, y0 + 10))
            gts.append(g)
            if r % 3 != 2:                       # a detection near this gt, shifted in time and space
                sh = int(rng.integers(-5, 6))
                ln2 = max(5, ln + int(rng.integers(-6, 7)))
                prs.append(dict(triplet=list(trip), score=float(rng.uniform(0.2, 1)), duration=[max(0, f0 + sh), max(0, f0 + sh) + ln2],
                                sub_traj=traj(ln2, x0 + rng.uniform(-8, 8), y0 + rng.uniform(-8, 8), 1.5),
                                obj_traj=traj(ln2, x0 + 90 + rng.uniform(-8, 8), y0 + 10 + rng.uniform(-8, 8), 1.5)))
        for r in range(6):                       # false positives
            ln = int(rng.integers(10, 30))
            f0 = int(rng.integers(0, 40))
            prs.append(dict(triplet=[names[int(rng.integers(0, 4))], preds_n[int(rng.integers(0, 3))], names[int(rng.integers(0, 4))]],
                            score=float(rng.uniform(0, 0.8)), duration=[f0, f0 + ln], sub_traj=traj(ln, 300, 300), obj_traj=traj(ln, 10, 300)))
        gt[vid], pr[vid] = gts, prs
    gt["empty_gt"] = []
    pr["vid03"] = pr["vid03"][:3]
    mean_ap, rec_at_n, mprec_at_n = V.evaluate(gt, pr)
    out["vid_gt"], out["vid_pred"] = jb(gt), jb(pr)
    out["vid_metrics"] = jb(dict(mean_ap=float(mean_ap), rec={str(k): float(v) for k, v in rec_at_n.items()},
                                 prec={str(k): float(v) for k, v in mprec_at_n.items()}))
    det = V.eval_detection_scores(gt["vid01"], pr["vid01"], 0.5)
    out["vid01_prec"], out["vid01_rec"], out["vid01_hit"] = det[0], det[1], det[2]
    print("videos: mAP %.4f, R@50 %.4f, P@1 %.4f" % (mean_ap, rec_at_n[50], mprec_at_n[1]))


def ckpt_maps(out):
    """lib/utils_rel/net_rel.py: the three name maps on every state-dict key of the model (Conv_Body, Prd_RCNN2D, RPN and
    the heads included) plus FPN-style names, and load_ckpt_rel on a small module (same-shape copy, skipped mismatch,
    the COCO -> AG class re-indexing branch)."""
    import torch
    import torch.nn as nn
    import utils_rel.net_rel as NR
    from trace_amd.model_builder_rel import Generalized_RCNN
    from trace_amd.rpn import single_scale_rpn_outputs
    m = Generalized_RCNN()
    m.RPN = single_scale_rpn_outputs(2048, 1 / 16)
    keys = list(m.state_dict().keys())
    keys += ["Conv_Body.conv_top.weight", "Conv_Body.topdown_lateral_modules.1.conv_lateral.bias", "Conv_Body.posthoc_modules.2.weight",
             "Conv_Body.conv_body.res1.conv1.weight", "Conv_Body.conv_body.res1.bn1.bias", "Conv_Body.conv_body.res3.2.conv2.weight",
             "Conv_Body.conv_body.res4.5.bn3.running_var", "Conv_Body.conv_body.res2.0.downsample.0.weight",
             "Conv_Body.conv_body.res2.0.downsample.1.bias", "RPN.FPN_RPN_conv.weight", "RPN.FPN_RPN_cls_score.bias",
             "RPN.FPN_RPN_bbox_pred.weight", "Box_Head.res5.0.bn2.weight", "Box_Head.res5.1.conv3.weight",
             "Box_Head.res5.0.downsample.0.weight", "Box_Head.res5.0.downsample.1.bias"]
    out["ckpt_keys"] = jb(keys)
    out["ckpt_res101"] = jb([NR.make_res101_map(k) for k in keys])
    out["ckpt_res50fpn"] = jb([NR.make_res50_fpn_map(k) for k in keys])
    out["ckpt_backbone"] = jb([NR.make_backbone_map(k) for k in keys])
    c2a, idx = NR.class_coco2ag()
    out["c2a"], out["c2a_idx"] = np.asarray(c2a, np.int64), np.asarray(idx, np.int64)

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
    rng = np.random.default_rng(4)
    ck = {"roi_heads.box_predictor.cls_score.weight": rng.normal(size=(81, 8)).astype(np.float32),
          "roi_heads.box_predictor.cls_score.bias": rng.normal(size=(81,)).astype(np.float32),
          "roi_heads.box_head.fc1.weight": rng.normal(size=(4, 8)).astype(np.float32),
          "roi_heads.box_head.fc1.bias": rng.normal(size=(5,)).astype(np.float32),           # shape mismatch: skipped
          "backbone.other.weight": rng.normal(size=(3, 3)).astype(np.float32)}
    with torch.no_grad():
        NR.load_ckpt_rel(t, ck)
    for k, v in ck.items():
        out["tiny_ck_" + k] = v
    for k, v in t.state_dict().items():
        out["tiny_sd_" + k] = v.numpy()
    print("ckpt maps: %d keys; tiny module loaded" % len(keys))


def main():
    cfg = ref_env.setup()
    out = {}
    frames(cfg, out)
    videos(out)
    ckpt_maps(out)
    np.savez_compressed(os.path.join(OUT, "py_eval.npz"), **out)
    print("py_eval.npz %d bytes" % os.path.getsize(os.path.join(OUT, "py_eval.npz")))


if __name__ == "__main__":
    main()
