"""Evaluation of relation outputs: per-frame triplet recall, video-level relation association and
the VidVRD detection / tagging metrics.  Host-side numpy, as in the reference (this is bookkeeping over a few hundred
triplets per frame, not device work).

Reference:
  per-frame top-k triplets + recall   lib/datasets_rel/task_evaluation_vg_and_vrd.py:159-330, `_compute_pred_matches`
                                      :366-406, `intersect_2d` / `argsort_desc` lib/datasets_rel/pytorch_misc.py:9-35,
                                      `boxes_union` lib/utils_rel/boxes_rel.py, `bbox_overlaps` utils/cython_bbox.pyx:32-73
  trajectories                        lib/vidvrd_utils/trajectory.py:11-157 (dlib.drectangle is a 4-float record here)
  greedy relational association       lib/vidvrd_utils/association.py:17-197
  detection mAP / recall / tagging    tools/transform_vidvrd_results.py:154-265, lib/vidvrd_utils/common.py:4-105
"""
from collections import defaultdict, deque
from copy import deepcopy
from functools import reduce

import numpy as np


# ----------------------------------------------------------------------------------------------- per-frame recall
def bbox_overlaps(boxes, query_boxes):
    """utils/cython_bbox.pyx:32-73: (N,4) x (K,4) -> (N,K) IoU with the +1 pixel convention, f32."""
    boxes = np.asarray(boxes, np.float32)
    query_boxes = np.asarray(query_boxes, np.float32)
    N, K = boxes.shape[0], query_boxes.shape[0]
    out = np.zeros((N, K), np.float32)
    if N == 0 or K == 0:
        return out
    # The compiled Cython evaluates "+ 1" and the area products / the union's sum in double (the literal becomes the C
    # double 1.0) and rounds to float32 when box_area, iw, ih, ua are assigned; differences, iw * ih and the division are
    # float32.  Bit for bit the module's own outputs (tests/golden/cy_overlaps.npz).
    f8 = np.float64
    d = lambda hi, lo: (hi - lo).astype(f8) + 1.0                                   # float32 difference, double "+ 1"
    qa = (d(query_boxes[:, 2], query_boxes[:, 0]) * d(query_boxes[:, 3], query_boxes[:, 1])).astype(np.float32)
    ba = d(boxes[:, 2], boxes[:, 0]) * d(boxes[:, 3], boxes[:, 1])                  # stays double inside `ua`
    iw = d(np.minimum(boxes[:, None, 2], query_boxes[None, :, 2]), np.maximum(boxes[:, None, 0], query_boxes[None, :, 0])).astype(np.float32)
    ih = d(np.minimum(boxes[:, None, 3], query_boxes[None, :, 3]), np.maximum(boxes[:, None, 1], query_boxes[None, :, 1])).astype(np.float32)
    ok = (iw > 0) & (ih > 0)
    inter = iw * ih                                                                 # float32
    ua = (ba[:, None] + qa[None, :].astype(f8) - inter.astype(f8)).astype(np.float32)
    out[ok] = inter[ok] / ua[ok]
    return out


def boxes_union(boxes1, boxes2):
    """lib/utils_rel/boxes_rel.py:42-50."""
    assert boxes1.shape == boxes2.shape
    return np.vstack((np.minimum(boxes1[:, 0], boxes2[:, 0]), np.minimum(boxes1[:, 1], boxes2[:, 1]),
                      np.maximum(boxes1[:, 2], boxes2[:, 2]), np.maximum(boxes1[:, 3], boxes2[:, 3]))).transpose()


def intersect_2d(x1, x2):
    """pytorch_misc.py:9-25: rows of x1 equal to rows of x2 -> (n1, n2) bool."""
    if x1.shape[1] != x2.shape[1]:
        raise ValueError("Input arrays must have same #columns")
    return (x1[..., None] == x2.T[None, ...]).all(1)


def argsort_desc(scores):
    """pytorch_misc.py:28-35."""
    return np.column_stack(np.unravel_index(np.argsort(-scores.ravel()), scores.shape))


def topk_frame_triplets(res, prd_k=1, topk=100, multi_relation=True):
    """task_evaluation_vg_and_vrd.py:159-253 for one frame's result dict (sbj_boxes, obj_boxes, sbj_labels, obj_labels,
    sbj_scores, obj_scores, prd_scores [, prd_scores_ttl])."""
    if res.get("prd_scores") is None:
        z4, zi = np.zeros((0, 4), np.float32), np.zeros(0, np.int32)
        return dict(det_boxes_s_top=z4, det_boxes_o_top=z4.copy(), det_labels_s_top=zi, det_labels_p_top=zi.copy(),
                    det_labels_o_top=zi.copy(), det_scores_top=np.zeros(0, np.float32))
    ss, os_ = res["sbj_scores"], res["obj_scores"]
    ss = ss if ss.ndim < 2 else np.amax(ss[:, 1:], axis=1)
    os_ = os_ if os_.ndim < 2 else np.amax(os_[:, 1:], axis=1)
    prd = res["prd_scores_ttl"] if "prd_scores_ttl" in res else res["prd_scores"]
    if not multi_relation:
        prd = prd[:, 1:]
    lab_prd = np.argsort(-prd, axis=1)
    sc_prd = -np.sort(-prd, axis=1)
    spo = (ss * os_)[:, None] * sc_prd[:, :prd_k]
    inds = argsort_desc(spo)[:topk]
    p_top = lab_prd[inds[:, 0], inds[:, 1]]
    return dict(det_boxes_s_top=res["sbj_boxes"][inds[:, 0]], det_boxes_o_top=res["obj_boxes"][inds[:, 0]],
                det_labels_s_top=res["sbj_labels"][inds[:, 0]], det_labels_p_top=p_top,
                det_labels_o_top=res["obj_labels"][inds[:, 0]], det_scores_top=spo[inds[:, 0], inds[:, 1]])


def compute_pred_matches(gt_triplets, pred_triplets, gt_boxes, pred_boxes, iou_thresh=0.5, phrdet=False):
    """task_evaluation_vg_and_vrd.py:366-406: for every prediction the list of gt relations it matches."""
    keeps = intersect_2d(gt_triplets, pred_triplets)
    gt_has_match = keeps.any(1)
    pred_to_gt = [[] for _ in range(pred_boxes.shape[0])]
    for gt_ind, gt_box, keep_inds in zip(np.where(gt_has_match)[0], gt_boxes[gt_has_match], keeps[gt_has_match]):
        boxes = pred_boxes[keep_inds].astype(np.float32, copy=False)
        gt_box = gt_box.astype(np.float32, copy=False)
        if phrdet:
            inds = bbox_overlaps(gt_box[None, :], boxes)[0] >= iou_thresh
        else:
            inds = (bbox_overlaps(gt_box[None, :4], boxes[:, :4])[0] >= iou_thresh) & \
                   (bbox_overlaps(gt_box[None, 4:], boxes[:, 4:])[0] >= iou_thresh)
        for i in np.where(keep_inds)[0][inds]:
            pred_to_gt[i].append(int(gt_ind))
    return pred_to_gt


def eval_frame_recall(all_results, topk=100, prd_k=1, recall_ks=(20, 50, 100), phrdet=False, multi_relation=True,
                      video_of=None):
    """task_evaluation_vg_and_vrd.py:118-330: image-level recall@k (matches summed over frames / gt relations) and
    video-level recall (per-video ratio, averaged).  all_results: list of per-frame dicts with the prediction keys of
    `topk_frame_triplets` plus gt_sbj_boxes, gt_obj_boxes, gt_sbj_labels, gt_obj_labels, gt_prd_labels and `image`.
    video_of(image) -> video key (default: the directory part of the image path, :136-141)."""
    if video_of is None:
        video_of = lambda name: name.rsplit("/", 1)[0] if "/" in name else name
    recalls = {k: 0 for k in recall_ks}
    video_recalls = defaultdict(lambda: {k: 0 for k in recall_ks})
    video_gt = defaultdict(int)
    all_gt = 0
    topk_dets = []
    for res in all_results:
        top = topk_frame_triplets(res, prd_k, topk, multi_relation)
        top["image"] = res["image"]
        gt_so = np.hstack((res["gt_sbj_boxes"], res["gt_obj_boxes"]))
        gt_spo = np.vstack((res["gt_sbj_labels"], res["gt_prd_labels"], res["gt_obj_labels"])).transpose()
        det_spo = np.vstack((top["det_labels_s_top"], top["det_labels_p_top"], top["det_labels_o_top"])).transpose()
        if phrdet:
            p2g = compute_pred_matches(gt_spo, det_spo, boxes_union(res["gt_sbj_boxes"], res["gt_obj_boxes"]),
                                       boxes_union(top["det_boxes_s_top"], top["det_boxes_o_top"]), 0.5, True)
        else:
            p2g = compute_pred_matches(gt_spo, det_spo, gt_so, np.hstack((top["det_boxes_s_top"], top["det_boxes_o_top"])), 0.5, False)
        vid = video_of(res["image"])
        all_gt += gt_spo.shape[0]
        video_gt[vid] += gt_spo.shape[0]
        for k in recall_ks:
            match = reduce(np.union1d, p2g[:k]) if len(p2g) else []
            recalls[k] += len(match)
            video_recalls[vid][k] += len(match)
        topk_dets.append(top)
    image_recall = {k: float(v) / (float(all_gt) + 1e-12) for k, v in recalls.items()}
    tot = {k: 0.0 for k in recall_ks}
    for vid, v in video_recalls.items():
        for k in recall_ks:
            tot[k] += float(v[k]) / (float(video_gt[vid]) + 1e-12)
    video_recall = {k: tot[k] / (float(len(video_recalls)) + 1e-12) for k in recall_ks}
    return image_recall, video_recall, topk_dets


# ----------------------------------------------------------------------------------------------- video association
class Trajectory:
    """lib/vidvrd_utils/trajectory.py:11-82 with boxes as (left, top, right, bottom) float tuples."""

    def __init__(self, pstart, pend, rois, score=0.0, category=0, classeme=(), vsig=None, gt_trackid=-1):
        assert len(rois) == pend - pstart
        self.pstart, self.pend = pstart, pend
        self.rois = deque(tuple(float(v) for v in roi) for roi in rois)
        self.score, self.category, self.classeme, self.vsig, self.gt_trackid = score, category, classeme, vsig, gt_trackid

    def length(self):
        return self.pend - self.pstart

    def predict(self, roi, reverse=False):
        if reverse:
            self.rois.appendleft(roi)
            self.pstart -= 1
        else:
            self.rois.append(roi)
            self.pend += 1
        return roi

    def serialize(self):
        return dict(pstart=int(self.pstart), pend=int(self.pend), rois=[tuple(r) for r in self.rois], score=float(self.score),
                    category=int(self.category), classeme=[float(x) for x in self.classeme], vsig=self.vsig,
                    gt_trackid=self.gt_trackid)


def cubic_iou(bboxes1, bboxes2):
    """trajectory.py:84-140: (n,t,4) x (m,t,4) -> (n,m) voluminal IoU, f32 accumulation as the reference."""
    b1 = np.asarray(bboxes1).transpose((1, 0, 2))
    b2 = np.asarray(bboxes2).transpose((1, 0, 2))
    inter = np.zeros((b1.shape[1], b2.shape[1]), np.float32)
    for i in range(b1.shape[0]):
        w = (np.minimum.outer(b1[i, :, 2], b2[i, :, 2]) + 1 - np.maximum.outer(b1[i, :, 0], b2[i, :, 0])).astype(np.float32).clip(min=0)
        h = (np.minimum.outer(b1[i, :, 3], b2[i, :, 3]) + 1 - np.maximum.outer(b1[i, :, 1], b2[i, :, 1])).astype(np.float32).clip(min=0)
        inter += w * h
    a1 = np.sum((b1[:, :, 2] - b1[:, :, 0] + 1) * (b1[:, :, 3] - b1[:, :, 1] + 1), axis=0)
    a2 = np.sum((b2[:, :, 2] - b2[:, :, 0] + 1) * (b2[:, :, 3] - b2[:, :, 1] + 1), axis=0)
    union = np.add.outer(a1, a2) - inter
    return inter / union


def _traj_iou(t1, t2):
    """association.py:38-51: IoU over the frames two trajectories share."""
    if t1.pend <= t2.pstart or t2.pend <= t1.pstart:
        return 0
    if t1.pstart > t2.pstart:
        t1, t2 = t2, t1
    r1 = list(t1.rois)[t2.pstart - t1.pstart:t1.pend - t1.pstart]
    r2 = list(t2.rois)[:t1.pend - t2.pstart]
    return cubic_iou(np.asarray([r1]), np.asarray([r2]))[0, 0]


def _merge_trajs(t1, t2):
    """association.py:17-35: average the overlapping frames, append the rest."""
    overlap = max(t1.pend - t2.pstart, 0)
    for i in range(overlap):
        a, b = t1.rois[t1.length() - overlap + i], t2.rois[i]
        t1.rois[t1.length() - overlap + i] = tuple((x + y) / 2 for x, y in zip(a, b))
    for i in range(overlap, t2.length()):
        t1.predict(t2.rois[i])
    return t1


class VideoRelation:
    """association.py:54-123."""

    def __init__(self, vid, s_cid, pid, o_cid, straj, otraj, confs=1):
        self.vid, self.s_cid, self.pid, self.o_cid, self.straj, self.otraj = vid, s_cid, pid, o_cid, straj, otraj
        self.confs_list = [confs]
        self.fstart, self.fend = straj.pstart, straj.pend

    def triplet(self):
        return (self.s_cid, self.pid, self.o_cid)

    def mean_confs(self):
        return np.max(self.confs_list)           # the reference's "mean" is the max (:87-89)

    def both_overlap(self, straj, otraj, iou_thr=0.5):
        miou = min(_traj_iou(self.straj, straj), _traj_iou(self.otraj, otraj))
        return miou >= iou_thr, miou

    def extend(self, straj, otraj, confs):
        self.straj = _merge_trajs(self.straj, straj)
        self.otraj = _merge_trajs(self.otraj, otraj)
        self.confs_list.append(confs)
        self.fstart, self.fend = self.straj.pstart, self.otraj.pend

    def serialize(self, obj_name_mapping, pred_name_mapping):
        return dict(triplet=[obj_name_mapping[self.s_cid], pred_name_mapping[self.pid], obj_name_mapping[self.o_cid]],
                    score=float(self.mean_confs()), duration=[int(self.fstart), int(self.fend)],
                    sub_traj=self.straj.serialize()["rois"], obj_traj=self.otraj.serialize()["rois"])


def greedy_relational_association(obj_name_mapping, pred_name_mapping, short_term_relations, trajectories_of,
                                  max_traj_num_in_clip=100):
    """association.py:126-197.  short_term_relations: list of ((vid, fstart, fend), [(score, (s, p, o), (s_traj_idx,
    o_traj_idx)), ...]); trajectories_of(vid, fstart, fend) -> list of Trajectory (the reference reads them from disk,
    trajectory.py:160-179)."""
    short_term_relations.sort(key=lambda x: int(x[0][1]))
    video_relation_list, last_modify = [], []
    for i, (index, pred_list) in enumerate(short_term_relations):
        vid, fstart, fend = index
        preds = sorted(pred_list, key=lambda x: x[0], reverse=True)[:max_traj_num_in_clip]
        trajs = trajectories_of(vid, fstart, fend)
        for t in trajs:
            t.pstart, t.pend = fstart, fend
        cur_modify = []
        if i > 0:
            last_modify.sort(key=lambda r: r.mean_confs(), reverse=False)
        for conf, (s_cid, pid, o_cid), (si, oi) in preds:
            straj, otraj = trajs[si], trajs[oi]
            match_r, match_miou = None, -1.
            if i > 0:
                for r in last_modify:
                    if (s_cid, pid, o_cid) == r.triplet() and straj.pstart < r.fend and otraj.pstart < r.fend:
                        ok, miou = r.both_overlap(straj, otraj)
                        if ok and match_miou <= miou:
                            match_miou, match_r = miou, r
            if match_r is None:
                r = VideoRelation(vid, s_cid, pid, o_cid, deepcopy(straj), deepcopy(otraj), confs=conf)
                video_relation_list.append(r)
                cur_modify.append(r)
            else:
                last_modify.remove(match_r)
                match_r.extend(straj, otraj, conf)
                cur_modify.append(match_r)
        last_modify = cur_modify
    return [r.serialize(obj_name_mapping, pred_name_mapping) for r in video_relation_list]


# ----------------------------------------------------------------------------------------------- VidVRD metrics
def voc_ap(rec, prec):
    """vidvrd_utils/common.py:4-37 (area under the precision envelope)."""
    mrec = np.concatenate(([0.], rec, [1.]))
    mpre = np.concatenate(([0.], prec, [0.]))
    for i in range(mpre.size - 1, 0, -1):
        mpre[i - 1] = np.maximum(mpre[i - 1], mpre[i])
    i = np.where(mrec[1:] != mrec[:-1])[0]
    return np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1])


def viou(traj_1, duration_1, traj_2, duration_2):
    """vidvrd_utils/common.py:65-105: voluminal IoU of two trajectories with durations [fstart, fend)."""
    if duration_1[0] >= duration_2[1] or duration_1[1] <= duration_2[0]:
        return 0.
    if duration_1[0] <= duration_2[0]:
        head_1, head_2 = duration_2[0] - duration_1[0], 0
    else:
        head_1, head_2 = 0, duration_1[0] - duration_2[0]
    if duration_1[1] < duration_2[1]:
        tail_1 = duration_1[1] - duration_1[0]
    else:
        tail_1 = duration_2[1] - duration_1[0]
    v_overlap = 0
    for i in range(tail_1 - head_1):
        a, b = traj_1[head_1 + i], traj_2[head_2 + i]
        v_overlap += max(0, min(a[2], b[2]) - max(a[0], b[0]) + 1) * max(0, min(a[3], b[3]) - max(a[1], b[1]) + 1)
    v1 = sum((r[2] - r[0] + 1) * (r[3] - r[1] + 1) for r in traj_1)
    v2 = sum((r[2] - r[0] + 1) * (r[3] - r[1] + 1) for r in traj_2)
    return float(v_overlap) / (v1 + v2 - v_overlap)


def eval_detection_scores(gt_relations, pred_relations, viou_threshold):
    """transform_vidvrd_results.py:154-182."""
    pred_relations = sorted(pred_relations, key=lambda x: x["score"], reverse=True)
    gt_detected = np.zeros((len(gt_relations),), dtype=bool)
    hit_scores = np.ones((len(pred_relations))) * -np.inf
    for pi, pr in enumerate(pred_relations):
        ov_max, k_max = -float("Inf"), -1
        for gi, gr in enumerate(gt_relations):
            if not gt_detected[gi] and tuple(pr["triplet"]) == tuple(gr["triplet"]):
                ov = min(viou(pr["sub_traj"], pr["duration"], gr["sub_traj"], gr["duration"]),
                         viou(pr["obj_traj"], pr["duration"], gr["obj_traj"], gr["duration"]))
                if ov >= viou_threshold and ov > ov_max:
                    ov_max, k_max = ov, gi
        if k_max >= 0:
            hit_scores[pi] = pr["score"]
            gt_detected[k_max] = True
    tp = np.isfinite(hit_scores)
    cum_tp = np.cumsum(tp).astype(np.float32)
    cum_fp = np.cumsum(~tp).astype(np.float32)
    rec = cum_tp / np.maximum(len(gt_relations), np.finfo(np.float32).eps)
    prec = cum_tp / np.maximum(cum_tp + cum_fp, np.finfo(np.float32).eps)
    return prec, rec, hit_scores


def eval_tagging_scores(gt_relations, pred_relations):
    """transform_vidvrd_results.py:185-206."""
    pred_relations = sorted(pred_relations, key=lambda x: x["score"], reverse=True)
    gt_triplets = set(tuple(r["triplet"]) for r in gt_relations)
    pred_triplets, hit_scores = [], []
    for r in pred_relations:
        t = tuple(r["triplet"])
        if t not in pred_triplets:
            pred_triplets.append(t)
            hit_scores.append(r["score"])
    hit_scores = np.asarray(hit_scores, dtype=np.float64)
    for i, t in enumerate(pred_triplets):
        if t not in gt_triplets:
            hit_scores[i] = -np.inf
    tp = np.isfinite(hit_scores)
    cum_tp = np.cumsum(tp).astype(np.float32)
    cum_fp = np.cumsum(~tp).astype(np.float32)
    rec = cum_tp / np.maximum(len(gt_triplets), np.finfo(np.float32).eps)
    prec = cum_tp / np.maximum(cum_tp + cum_fp, np.finfo(np.float32).eps)
    return prec, rec, hit_scores


def evaluate(groundtruth, prediction, viou_threshold=0.5, det_nreturns=(50, 100), tag_nreturns=(1, 5, 10)):
    """transform_vidvrd_results.py:209-265: (detection mean AP, detection recall@n, tagging precision@n)."""
    video_ap = {}
    tot_scores, tot_tp, prec_at_n = defaultdict(list), defaultdict(list), defaultdict(list)
    tot_gt = 0
    for vid, gt_relations in groundtruth.items():
        if len(gt_relations) == 0:
            continue
        tot_gt += len(gt_relations)
        preds = prediction.get(vid, [])
        det_prec, det_rec, det_scores = eval_detection_scores(gt_relations, preds, viou_threshold)
        video_ap[vid] = voc_ap(det_rec, det_prec)
        tp = np.isfinite(det_scores)
        for n in det_nreturns:
            cut = min(n, det_scores.size)
            tot_scores[n].append(det_scores[:cut])
            tot_tp[n].append(tp[:cut])
        tag_prec, _, _ = eval_tagging_scores(gt_relations, preds)
        for n in tag_nreturns:
            cut = min(n, tag_prec.size)
            prec_at_n[n].append(tag_prec[cut - 1] if cut > 0 else 0.)
    mean_ap = np.mean(list(video_ap.values()))
    rec_at_n = {}
    for n in det_nreturns:
        scores, tps = np.concatenate(tot_scores[n]), np.concatenate(tot_tp[n])
        tps = tps[np.argsort(scores)[::-1]]
        rec_at_n[n] = (np.cumsum(tps).astype(np.float32) / np.maximum(tot_gt, np.finfo(np.float32).eps))[-1]
    return mean_ap, rec_at_n, {n: np.mean(prec_at_n[n]) for n in tag_nreturns}
