"""TEST INFRASTRUCTURE ONLY — fp32 CPU restatement of the TRACE relation head for ONE
clip (the reference's effective batch size, lib/modeling_rel/relpn_heads.py:67,79-80),
driven by a plain state_dict with the reference's parameter names.

Follows lib/modeling_rel/model_builder_rel.py `_forward` :664-1204 in the inference /
pre-computed-box flow (:704-731, :889-907, :938-952, :966-1088) with the VidVRD yaml
(configs/vidvrd/vidvrd_res101xi3d50_gt_boxes_dc5_2d_new.yaml: NO_FC7_RELU, MULTI_RELATION,
USE_FREQ_BIAS, I3D_DC5, RESIZE_HIDDEN_DIM=4), lib/combine_tree/cluster_tree3d.py:114-366,
lib/combine_tree/cluster_tree.py:163-352, lib/i3d/box_head3d.py:53-79 and
lib/modeling_rel/fast_rcnn_heads.py:15-51,94-152.

The three 2-D / 3-D conv bodies are outside this function: their outputs (blob_conv,
feat_map, temporal maps) are inputs.

Pinning: the stages below are checked against outputs of the reference's own modules run on the CPU
of the build container (tools/make_goldens_py.py -> tests/golden/py_heads.npz, py_cluster_tree3d.npz,
py_pair_head.npz; tests/test_oracle_goldens_py.py): roi_2mlp_head / fast_rcnn_outputs / Tail_Layer_3D,
the whole cluster_tree3d.forward (temporal ROIAlign order, attention, region tree, tree-GRU, FFN),
get_spatio_atten_feat and get_sem_feat.  ROIAlign inside is the pinned C oracle.  NOT pinned by
execution (Generalized_RCNN() itself cannot be constructed here: mmcv, GloVe and the dataset
statistics are absent): the glue of `_forward` between those stages (:938-952, :1018-1024,
:1039-1042, :1064-1083), restated line by line.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import c_oracle, np_oracle


def _lin(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))


def _roi_align(fmap, rois, scale, sr):
    out = c_oracle.roi_align_fwd(fmap.numpy(), np.ascontiguousarray(rois, np.float32), scale, 7, 7, sr)
    return torch.from_numpy(out)


def _two_mlp(sd, prefix, pooled, use_relu):
    """roi_2mlp_head.map2vec, fast_rcnn_heads.py:140-152."""
    x = F.relu(_lin(sd, prefix + ".fc1", pooled.reshape(pooled.shape[0], -1)))
    x = _lin(sd, prefix + ".fc2", x)
    return F.relu(x) if use_relu else x


def _tail(sd, prefix, pooled):
    """Tail_Layer_3D.map2vec with I3D_DC5 (box_head3d.py:59-68): avg-pool 7x7 -> fc2 -> ReLU."""
    x = pooled.mean(dim=(2, 3))
    return F.relu(_lin(sd, prefix + ".fc2", x))


def _temporal_attention(sd, prefix, s, o, low_dim):
    """cluster_tree3d.py:337-366."""
    q_s = _lin(sd, prefix + "_q_s.0", s).unsqueeze(1)
    q_o = _lin(sd, prefix + "_q_o.0", o)
    score = torch.mul(q_s, q_o).sum(2) / math.sqrt(low_dim)
    score = F.softmax(score, 1).unsqueeze(2)
    return (score * o).sum(1)


def _feat_val(sd, prefix, x):
    """cluster_tree3d.py:80-84 / 91-95: Linear -> LayerNorm -> ReLU -> Linear (no biases on the Linears)."""
    x = _lin(sd, prefix + ".0", x)
    x = F.layer_norm(x, (x.shape[-1],), sd[prefix + ".1.weight"], sd[prefix + ".1.bias"], 1e-5)
    return _lin(sd, prefix + ".3", F.relu(x))


def temporal_rois(boxes, T, n):
    """The roi list the reference feeds the temporal ROIAlign (cluster_tree3d.py:158-162, 178-182):
    box list is object-major (static tubes, model_builder_rel.py:1441-1444, reshape(-1,4)) while the
    frame ids are frame-major; the pooled result is then viewed (T, n, ...).  So entry (t, j) pairs
    frame t with the box of object (t*n + j) // T.  Reproduced as is."""
    flat_boxes = np.repeat(boxes, T, axis=0)                      # object-major
    ids = np.repeat(np.arange(T), n).astype(np.float32)[:, None]   # frame-major
    return np.concatenate([ids, flat_boxes], 1).astype(np.float32)


def _tree_gru(sd, prefix, feat, layers, n_tree, A, A_T):
    """tree_message_passing2D.forward_v1, cluster_tree.py:163-204, 281-330 (dense masks as there)."""
    Hd = sd[prefix + ".Wh.0.weight"].shape[0]
    W = lambda n, x: _lin(sd, prefix + "." + n + ".0", x)
    h = torch.zeros((n_tree, Hd))
    fa_sets = [list(l.keys()) for l in layers]
    son_sets = [np.hstack([np.array(v) for v in l.values()]) for l in layers]
    A = torch.from_numpy(A)
    A_T = torch.from_numpy(A_T)
    x0 = feat[son_sets[0]]
    z = torch.sigmoid(W("Wz", x0))
    h[son_sets[0]] = (1 - z) * torch.tanh(W("Wh", x0))
    for l in range(len(layers)):
        fa, son = fa_sets[l], son_sets[l]
        x = feat[fa]
        hs = h[son]
        cur = A_T[fa][:, son]
        h_sum = cur @ hs
        z = torch.sigmoid(W("Wz", x) + W("Uz", h_sum))
        part = torch.sigmoid(W("Wr", x).unsqueeze(1) + W("Ur", hs).unsqueeze(0))
        r = cur.unsqueeze(2) * part
        fa_g = W("Uh", (r * hs.unsqueeze(0)).sum(1))
        h[fa] = z * h_sum + (1 - z) * torch.tanh(W("Wh", x) + fa_g)
    hd = torch.zeros((n_tree, Hd))
    hd[fa_sets[-1]] = h[fa_sets[-1]]
    for l in range(len(layers) - 1, -1, -1):
        fa, son = fa_sets[l], son_sets[l]
        cur = A[son][:, fa]
        s_i, f_i = torch.where(cur > 0)
        x = feat[son]
        hf = hd[fa][f_i]
        z = torch.sigmoid(W("Wzd", x) + W("Uzd", hf))
        r = torch.sigmoid(W("Wrd", x) + W("Urd", hf))
        hd[son] = z * hf + (1 - z) * torch.tanh(W("Whd", x) + W("Uhd", r * hf))
    return _lin(sd, prefix + ".FFN1.0", h) + _lin(sd, prefix + ".FFN2.0", hd)


def cluster_tree3d_forward(sd, prefix, roi, feat_map, obj_feat_vec, tmaps, low_dim=256, groups=4,
                           max_tree_height=7, partition=2, rel_scale=1.0 / 16):
    """cluster_tree3d.forward for one clip, lib/combine_tree/cluster_tree3d.py:114-335.
      roi (n,5) f32 [0,x1,y1,x2,y2]; feat_map (1,C,h,w); obj_feat_vec (n,K); tmaps (T',C,h,w); static tubes
      (model_builder_rel.py:1441-1444: temporal_roi = roi repeated over T')."""
    P = prefix + "."
    n = roi.shape[0]
    Tn = tmaps.shape[0]
    boxes = np.ascontiguousarray(roi[:, 1:5], np.float32)
    t_rois = temporal_rois(boxes, Tn, n)
    t_pool = _roi_align(tmaps, t_rois, rel_scale, 0)                      # :169-173
    t_feat = _tail(sd, P + "temporal_obj_fc", t_pool)                     # :186
    t_feat = t_feat.view(Tn, n, -1).permute(1, 0, 2).contiguous()         # :181-182,187
    att = _temporal_attention(sd, P + "temporal_obj_atten", obj_feat_vec, t_feat, low_dim)  # :192
    obj_agg = obj_feat_vec + _feat_val(sd, P + "temporal_obj_feat_val", att)  # :195,201
    layers, tree = np_oracle.node_cluster(boxes, max_tree_height, partition)  # :205-214
    n_tree = tree.shape[0]
    m = n_tree - n
    A, A_T, link = np_oracle.get_A_and_link(n_tree, layers)               # :219
    reg_boxes = tree[n:].astype(np.float32)
    reg_rois = np.concatenate([np.zeros((m, 1), np.float32), reg_boxes], 1)
    reg_map = _roi_align(feat_map, reg_rois, rel_scale, 0)                # :238-242
    reg_feat = _lin(sd, P + "fc_r.2", F.relu(_lin(sd, P + "fc_r.0", reg_map.reshape(m, -1))))  # :244
    t_boxes = np.repeat(boxes[:, None, :], Tn, axis=1)
    t_reg = np_oracle.temporal_nonleaf_rois(layers, n_tree, n, t_boxes)   # :247-251 (m,T,4)
    tr_rois = np.concatenate([np.repeat(np.arange(Tn), m).astype(np.float32)[:, None], t_reg.reshape(-1, 4)], 1)
    tr_pool = _roi_align(tmaps, tr_rois, rel_scale, 0)                    # :268-272
    tr_feat = _tail(sd, P + "temporal_r_fc", tr_pool).view(Tn, m, -1).permute(1, 0, 2).contiguous()
    att_r = _temporal_attention(sd, P + "temporal_r_atten", reg_feat, tr_feat, low_dim)  # :293
    reg_agg = reg_feat + _feat_val(sd, P + "temporal_r_feat_val", att_r)  # :297,305
    r2c = _lin(sd, P + "r2c.1", F.relu(reg_agg))                          # :307
    o2c = _lin(sd, P + "o2c.1", F.relu(obj_agg))                          # :308
    pooled_features = torch.cat([o2c, r2c], 0)                            # :310-316
    gw = pooled_features.shape[1] // groups
    agg = [_tree_gru(sd, P + "tree_message_passing_list.%d" % g, pooled_features[:, gw * g:gw * (g + 1)],
                     layers, n_tree, A, A_T) for g in range(groups)]      # :319-327
    pooled_features = _lin(sd, P + "FFN.1", F.relu(torch.cat(agg, 1)))    # :330
    return dict(pooled_features=pooled_features, region_map=reg_map, layers=layers, tree_boxes=tree, A=A, A_T=A_T,
                link=link, obj_agg=obj_agg, reg_agg=reg_agg, nonleaf_rois=reg_rois)


def pair_attention(sd, roi, feat_map, obj_feature, link, sbj_inds, obj_inds, rel_scale=1.0 / 16):
    """get_spatio_atten_feat for one clip, lib/modeling_rel/model_builder_rel.py:1524-1612.
    Returns s_map, o_map, v_map (P,K,7,7), lca as REGION index (:1559), union rois, q_s, q_o."""
    n = roi.shape[0]
    q_s = _lin(sd, "spatio_atten_s.1", F.leaky_relu(obj_feature, 0.1))    # :1540
    q_o = _lin(sd, "spatio_atten_o.1", F.leaky_relu(obj_feature, 0.1))    # :1541
    v_full = F.conv2d(feat_map, sd["spatio_atten_map.weight"], sd["spatio_atten_map.bias"])  # :1542
    k_full = F.conv2d(feat_map, sd["atten_q_map.weight"], sd["atten_q_map.bias"])            # :1543
    lca = np_oracle.get_lca(sbj_inds, obj_inds, link) - n                 # :1553,1559
    union = np_oracle.rois_union(roi[sbj_inds], roi[obj_inds]).astype(np.float32)  # :1561-1562,1588-1594
    k_p = _roi_align(k_full, union, rel_scale, 0)                         # :1566
    v_p = _roi_align(v_full, union, rel_scale, 0)                         # :1568

    def mask_map(q):                                                      # :1596-1612
        score = (q.view(q.shape[0], q.shape[1], 1, 1) * k_p).sum(1).view(q.shape[0], -1)
        score = F.softmax(score, 1).view(k_p.shape[0], 1, k_p.shape[2], k_p.shape[3])
        return v_p * score
    return mask_map(q_s[sbj_inds]), mask_map(q_o[obj_inds]), v_p, lca, union, q_s, q_o


def sem_feat(sd, softmax_obj_vec, region_feature, lca, sbj_inds, obj_inds):
    """get_sem_feat for one clip, model_builder_rel.py:1643-1668."""
    low_s = _lin(sd, "sem_s_e.1", F.leaky_relu(_lin(sd, "embed_s", softmax_obj_vec), 0.1))
    low_o = _lin(sd, "sem_o_e.1", F.leaky_relu(_lin(sd, "embed_o", softmax_obj_vec), 0.1))
    low_r = _lin(sd, "sem_r.1", F.leaky_relu(region_feature, 0.1))
    return low_s[sbj_inds] + low_r[lca] + low_o[obj_inds]


def head_forward(sd, blob_conv, feat_map, tmaps, det_rois, im_scale=1.0, f_scale=1.0, groups=4,
                 low_dim=256, max_tree_height=7, partition=2, conv_scale=1.0 / 16, rel_scale=1.0 / 16,
                 det_labels_in=None, det_scores_in=None, dataset_name="vidvrd", use_ovlp_filter=False,
                 pairs=None, pair_labels=None):
    """One clip.
      blob_conv (1,2048,H,W)  Conv_Body output (model_builder_rel.py:702)
      feat_map  (1,2048,h,w)  Prd_RCNN2D output (:1006-1007)
      tmaps     (T',2048,h,w) I3D output frames (:978-981)
      det_rois  (n,5) f32 [0,x1,y1,x2,y2] in the scaled key-frame (:894)
      pairs = (sbj_inds, obj_inds): the gt-box branch (:772-888), where det_rois are the de-duplicated gt boxes and
              the pairs are the annotated ones (np_oracle.gt_box_pairs); Box_Head / S_Head / O_Head pool sbj_rois /
              obj_rois there (:824-829, :887-888), which are det_rois[sbj_inds] / det_rois[obj_inds] row for row
      pair_labels = dict(sbj_labels, obj_labels [1-based], sbj_scores, obj_scores) per pair: use_gt_labels (:844-858)
    Returns dict with prd_scores (P,n_prd), obj_scores (n,C), det_labels, det_scores, sbj_inds,
    obj_inds, lca and intermediate features."""
    sd = {k: v.float().cpu() for k, v in sd.items()}
    n = det_rois.shape[0]
    Tn = tmaps.shape[0]
    # ---- detection heads on the key frame (:742-743, :900-904, :938-952)
    pooled = _roi_align(blob_conv, det_rois, conv_scale, 2)
    ori = _two_mlp(sd, "Box_Head", pooled, use_relu=False)               # :938
    cls = F.softmax(_lin(sd, "Box_Outs.cls_score", F.relu(ori)), dim=1)   # :939 (= :743 on the same rois)
    cls_np = cls.numpy()
    det_labels = np.argmax(cls_np[:, 1:], axis=1).astype(np.int32) + 1    # :903
    det_scores = np.amax(cls_np[:, 1:], axis=1)                           # :904
    if det_labels_in is not None:                                         # `prd_category` records (:895-898) /
        det_labels = np.asarray(det_labels_in).astype(np.int32)           # prepare_det_rois output (:912-915)
        det_scores = np.asarray(det_scores_in).astype(np.float32)
    if pairs is not None:                                                 # gt-box branch (:802-810)
        sbj_inds, obj_inds = np.asarray(pairs[0]).astype(np.int64), np.asarray(pairs[1]).astype(np.int64)
        rel_rois = np_oracle.rois_union(det_rois[sbj_inds], det_rois[obj_inds])
    else:
        sbj_inds, obj_inds, rel_rois = np_oracle.relpn_pairs(det_rois, det_labels if det_labels_in is not None else None,
                                                             dataset_name)   # RelPN :906
    if use_ovlp_filter and pairs is None:                                                   # relpn_heads.py:94-105
        sb = (det_rois[sbj_inds][:, 1:] / im_scale).astype(np.float32)
        ob = (det_rois[obj_inds][:, 1:] / im_scale).astype(np.float32)
        ovlp = c_oracle.bbox_pair_overlaps(sb, ob)                        # cython_bbox_rel.pyx:37-77
        keep = np.where(ovlp > 0)[0]
        if keep.size > 0:
            sbj_inds, obj_inds, rel_rois = sbj_inds[keep], obj_inds[keep], rel_rois[keep]
    det_embed = _lin(sd, "embed_obj_det", cls)                            # :945
    det_feat = ori + _lin(sd, "after_obj_det_embed.1", F.leaky_relu(det_embed, 0.1))  # :946
    det_s = _two_mlp(sd, "S_Head", pooled, use_relu=False)                # :949 (NO_FC7_RELU)
    det_o = _two_mlp(sd, "O_Head", pooled, use_relu=False)
    sbj_feat, obj_feat = det_s[sbj_inds], det_o[obj_inds]                 # :951-952
    # ---- tubes + down-scaling (:966-975, :1766-1775)
    roi = det_rois.copy()
    roi[:, 1:5] *= float(1.0 * f_scale * (1.0 / im_scale))
    roi = roi.astype(np.float32)
    boxes = roi[:, 1:5]
    ct = cluster_tree3d_forward(sd, "cluster_tree3d", roi, feat_map, det_feat, tmaps, low_dim=low_dim, groups=groups,
                                max_tree_height=max_tree_height, partition=partition, rel_scale=rel_scale)
    pooled_features, layers, tree, link = ct["pooled_features"], ct["layers"], ct["tree_boxes"], ct["link"]
    obj_agg, reg_agg = ct["obj_agg"], ct["reg_agg"]
    # ---- back in _forward (:1018-1024)
    obj_feature = det_feat + _lin(sd, "restore_obj_feat.1", F.leaky_relu(pooled_features[:n], 0.1))
    region_feature = pooled_features[n:]
    obj_score = F.softmax(_lin(sd, "cem_Box_Outs.cls_score", F.relu(obj_feature)), dim=1)  # :1024,1084
    s_map, o_map, v_p, lca, union, q_s, q_o = pair_attention(sd, roi, feat_map, obj_feature, link, sbj_inds, obj_inds,
                                                             rel_scale=rel_scale)
    sp = F.leaky_relu(F.conv2d(s_map, sd["spatio_inferA.0.weight"], sd["spatio_inferA.0.bias"]) +
                      F.conv2d(o_map, sd["spatio_inferB.0.weight"], sd["spatio_inferB.0.bias"]), 0.1)  # :1039
    sp = (sp + v_p).view(sp.shape[0], -1)                                 # :1040-1041
    spatio_infer = _lin(sd, "spatio_infer_fc.0", sp)                      # :1042
    sem = sem_feat(sd, cls, region_feature, lca, sbj_inds, obj_inds)      # :1044
    # ---- scoring (:1064-1083)
    if pair_labels is not None:                                           # all_sbj_labels_int32 (:875-876, :1051-1052)
        s_lab = torch.from_numpy(np.asarray(pair_labels["sbj_labels"]).astype(np.int64) - 1)
        o_lab = torch.from_numpy(np.asarray(pair_labels["obj_labels"]).astype(np.int64) - 1)
    else:
        s_lab = torch.from_numpy(det_labels[sbj_inds].astype(np.int64) - 1)
        o_lab = torch.from_numpy(det_labels[obj_inds].astype(np.int64) - 1)
    tab = sd["freq_bias.rel_baseline.weight"]
    n_obj = int(round(math.sqrt(tab.shape[0])))
    bias = tab[s_lab * n_obj + o_lab]                                     # sparse_targets_rel.py:67-72
    logits = (_lin(sd, "relation_infer_spa.2", F.leaky_relu(spatio_infer, 0.1)) +
              _lin(sd, "relation_infer_sem.2", F.leaky_relu(sem, 0.1)) +
              _lin(sd, "sem_s.0", sbj_feat) + _lin(sd, "sem_o.0", obj_feat)) + bias
    prd = torch.sigmoid(logits)                                           # :1083 (MULTI_RELATION)
    return dict(prd_scores=prd, logits=logits, obj_scores=obj_score, det_labels=det_labels, det_scores=det_scores,
                sbj_inds=sbj_inds, obj_inds=obj_inds, lca=lca, det_feat=det_feat, obj_agg=obj_agg,
                reg_agg=reg_agg, pooled_features=pooled_features, obj_feature=obj_feature, q_s=q_s, q_o=q_o,
                s_map=s_map, o_map=o_map, v_p=v_p, spatio_infer=spatio_infer, sem=sem, sbj_feat=sbj_feat,
                obj_feat=obj_feat, tree_boxes=tree, layers=layers, rel_rois=union, roi=roi)
