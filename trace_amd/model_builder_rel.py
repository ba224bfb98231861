"""TRACE relation model on MI355X — the per-clip hot path of
lib/modeling_rel/model_builder_rel.py `Generalized_RCNN` (:107-360 modules, :664-1204 forward,
inference / pre-computed-box flow) with lib/combine_tree/cluster_tree3d.py:114-366.

Same sub-module names and parameter shapes as the reference (Box_Head, Box_Outs, cem_Box_Outs,
S_Head, O_Head, Prd_RCNN, cluster_tree3d, embed_*, spatio_*, sem_*, relation_infer_*,
freq_bias.rel_baseline, Conv_Body, Prd_RCNN2D) so its checkpoints load.  `forward_clips` takes the
2-D bodies' outputs `blob_conv` / `feat_map` as inputs (the batched entry); `forward` runs the bodies
itself like the reference.

MI355X-first differences from the reference's control flow (results are the same):
  * B clips are processed together (the reference is 1 clip per GPU call): one I3D pass, one
    ROIAlign per feature map, one GEMM per layer over all clips' rois / pairs;
  * activations are channels-last bf16, every Linear / 1x1 conv is the bf16 MFMA GEMM of
    libtrace_amd.so, ROIAlign / temporal attention / pair attention / tree-GRU are HIP kernels;
  * Box_Head, S_Head and O_Head pool the SAME rois from the SAME map (:938,:949-950): pooled once,
    the three fc1 run as one GEMM; `Box_Head(use_relu=True)` + Box_Outs (:742-743) equals
    Box_Outs(relu(Box_Head(use_relu=False))) (:938-939) on the pre-computed rois: computed once;
  * the box tree is built on the host (as in the reference) while the I3D kernels run; it is
    uploaded once as parent / CSR arrays instead of 8 dense adjacency matrices per clip.
Small elementwise glue (softmax over 36 classes, LayerNorm(256), gathers, sigmoid) is torch on
the GPU; it is listed in DESIGN.md.
"""
import collections
import gc
import math
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import boxes as box_ops
from . import cluster_tree as ct
from . import ops
from . import resnet2d
from .config import TraceConfig, vidvrd_config
from .heads import (PackedLinears, cluster_tree3d, fast_rcnn_outputs, ffn12_weights, init_msra_conv, init_xavier,
                    roi_2mlp_head, tree_gru_weights, xavier_fill)
from .roi_xform import add_rel_multilevel_rois, roi_feature_transform
from .resnet3d import _PACK_EPOCH, ResNet3d, i3d_dc5_config


class FrequencyBias(nn.Module):
    """lib/modeling_rel/sparse_targets_rel.py:23-72 (table supplied by the caller / checkpoint)."""

    def __init__(self, num_objs, num_prd):
        super().__init__()
        self.num_objs = num_objs
        self.rel_baseline = nn.Embedding(num_objs * num_objs, num_prd)

    def rel_index_with_labels(self, labels):
        return self.rel_baseline(labels[:, 0] * self.num_objs + labels[:, 1])


def _bf(x):
    return x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16)


def swapped_pair_index(X, Y):
    """partner[p] = q with (X[q], Y[q]) == (Y[p], X[p]), -1 when the swapped pair is not in the list (person-only
    subjects, overlap filter) or is p itself.  rois_union (boxes_rel.py:100-106) is symmetric, so p and q pool the same
    union box: `ops.pair_infer` pools it once for both."""
    X = np.asarray(X, np.int64)
    Y = np.asarray(Y, np.int64)
    if X.size == 0:
        return np.zeros(0, np.int32)
    base = int(max(X.max(), Y.max())) + 1
    key = X * base + Y
    order = np.argsort(key, kind="stable")
    skey = key[order]
    want = Y * base + X
    pos = np.minimum(np.searchsorted(skey, want), len(skey) - 1)
    q = order[pos]
    ok = (skey[pos] == want) & (q != np.arange(len(X)))
    part = np.where(ok, q, -1)
    # keep only mutual, unique matches (duplicate pairs in a list would break the one-writer rule)
    safe = part >= 0
    safe[safe] = part[part[safe]] == np.nonzero(safe)[0]
    return np.where(safe, part, -1).astype(np.int32)


class _Upload:
    """All per-step host arrays (roi lists, forest CSR arrays, pair indices) travel in ONE pinned
    staging buffer and one stream-ordered copy: the reference uploads each list with a blocking
    `torch.from_numpy(..).cuda()` (model_builder_rel.py:1389, cluster_tree.py:214,730)."""

    _DT = {np.dtype(np.float32): torch.float32, np.dtype(np.int32): torch.int32, np.dtype(np.int64): torch.int64}

    def __init__(self):
        self.items = []
        self.nbytes = 0

    def add(self, a):
        a = np.ascontiguousarray(a)
        if a.dtype not in self._DT:
            raise TypeError("upload: unsupported dtype %s" % a.dtype)
        self.items.append((self.nbytes, a))
        self.nbytes = (self.nbytes + a.nbytes + 255) // 256 * 256
        return len(self.items) - 1

    def commit(self, dev, into=None):
        """Copy to the device (stream-ordered, pinned staging) and return the per-array views; `into` is
        an existing device buffer of the same layout (the resident buffer of a captured step)."""
        host = torch.empty((max(self.nbytes, 256),), dtype=torch.uint8, pin_memory=True)
        hv = host.numpy()
        for off, a in self.items:
            hv[off:off + a.nbytes] = a.reshape(-1).view(np.uint8)
        if into is None:
            d = host.to(dev, non_blocking=True)
        else:
            if into.numel() != host.numel():
                raise RuntimeError("upload layout changed under a captured step")
            into.copy_(host, non_blocking=True)
            d = into
        return [d[off:off + a.nbytes].view(self._DT[a.dtype]).view(a.shape) for off, a in self.items]

    def views(self, d):
        """Per-array views of a device buffer laid out like this upload."""
        return [d[off:off + a.nbytes].view(self._DT[a.dtype]).view(a.shape) for off, a in self.items]

    def fill(self, host_u8):
        """Pack the arrays into a (pinned) host uint8 numpy view of at least `nbytes`."""
        for off, a in self.items:
            host_u8[off:off + a.nbytes] = a.reshape(-1).view(np.uint8)


class Generalized_RCNN(PackedLinears):
    def __init__(self, cfg: TraceConfig = None, conv_bodies=True):
        super().__init__()
        cfg = cfg or vidvrd_config()
        self.cfg = cfg
        C = cfg.CONV_BODY_DIM
        D = cfg.MLP_HEAD_DIM
        low = cfg.LOW_FEAT_DIM
        n_rel = cfg.NUM_PRD_CLASSES                      # MULTI_RELATION (:157-158)
        infer = D // cfg.RESIZE_HIDDEN_DIM
        fc = D // cfg.RESIZE_HIDDEN_DIM
        E = cfg.OBJ_EMBED_DIM
        ncls = cfg.NUM_CLASSES
        self.pool_size = cfg.ROI_XFORM_RESOLUTION
        self.Box_Head = roi_2mlp_head(C, cfg)
        self.Box_Outs = fast_rcnn_outputs(D, cfg)
        self.Prd_RCNN = ResNet3d(**i3d_dc5_config())      # :166-183
        self.cluster_tree3d = cluster_tree3d(self.Prd_RCNN.feat_dim, cfg)   # :238-251
        self.embed_obj_det = nn.Linear(ncls, E, bias=False)
        self.embed_s = nn.Linear(ncls, E, bias=False)
        self.embed_o = nn.Linear(ncls, E, bias=False)
        self.after_obj_det_embed = nn.Sequential(nn.LeakyReLU(0.1), nn.Linear(E, D))
        self.restore_obj_feat = nn.Sequential(nn.LeakyReLU(0.1), nn.Linear(fc, D, bias=True))
        self.spatio_atten_s = nn.Sequential(nn.LeakyReLU(0.1), nn.Linear(D, low, bias=True))
        self.spatio_atten_o = nn.Sequential(nn.LeakyReLU(0.1), nn.Linear(D, low, bias=True))
        self.spatio_atten_map = nn.Conv2d(self.Prd_RCNN.feat_dim, low, 1, bias=True)
        self.atten_q_map = nn.Conv2d(self.Prd_RCNN.feat_dim, low, 1, bias=True)
        self.spatio_inferA = nn.Sequential(nn.Conv2d(low, low, 1, bias=True))
        self.spatio_inferB = nn.Sequential(nn.Conv2d(low, low, 1, bias=True))
        self.spatio_infer_fc = nn.Sequential(nn.Linear(low * self.pool_size ** 2, infer, bias=True))
        self.sem_s = nn.Sequential(nn.Linear(D, n_rel, bias=True))
        self.sem_s_e = nn.Sequential(nn.LeakyReLU(0.1), nn.Linear(E, infer, bias=True))
        self.sem_o = nn.Sequential(nn.Linear(D, n_rel, bias=True))
        self.sem_o_e = nn.Sequential(nn.LeakyReLU(0.1), nn.Linear(E, infer, bias=True))
        self.sem_r = nn.Sequential(nn.LeakyReLU(0.1), nn.Linear(fc, infer, bias=True))
        self.relation_infer_sem = nn.Sequential(nn.Dropout(p=0.5), nn.LeakyReLU(0.1, inplace=True),
                                                nn.Linear(infer, n_rel, bias=True))
        self.relation_infer_spa = nn.Sequential(nn.Dropout(p=0.5), nn.LeakyReLU(0.1, inplace=True),
                                                nn.Linear(infer, n_rel, bias=True))
        self.freq_bias = FrequencyBias(ncls - 1, n_rel)
        # :362-391 reset()
        for m in (self.after_obj_det_embed, self.restore_obj_feat, self.spatio_atten_s, self.spatio_atten_o,
                  self.spatio_infer_fc, self.sem_s, self.sem_s_e, self.sem_o, self.sem_o_e, self.sem_r,
                  self.relation_infer_sem, self.relation_infer_spa):
            m.apply(init_xavier)
        init_xavier(self.spatio_atten_map)
        init_xavier(self.atten_q_map)
        self.spatio_inferA.apply(init_msra_conv)
        self.spatio_inferB.apply(init_msra_conv)
        # :349-350, :419-420
        import copy
        self.S_Head = copy.deepcopy(self.Box_Head)
        self.O_Head = copy.deepcopy(self.S_Head)
        self.cem_Box_Outs = fast_rcnn_outputs(D, cfg, with_bbox=False)
        self.cem_Box_Outs.load_state_dict({k: v for k, v in self.Box_Outs.state_dict().items() if "cls_score" in k})
        self._pk = {}
        # replay the I3D launches from a HIP graph (one graph per clip-batch shape); serving loops with a
        # fixed batch shape switch this on, shape-varying callers leave it off
        self.graph_backbone = False
        # replay the WHOLE step (backbone + head) from one HIP graph per step signature
        self.graph_step = False
        # captured graphs, least recently used first; at most `max_step_graphs` step signatures stay captured per mode
        # (a graph holds its own memory pool: a shape-varying caller would otherwise grow without bound) - `_evict`
        self._step_graphs = collections.OrderedDict()
        self.max_step_graphs = 8
        self._last_step_entry = None
        # graph_step only: a call captures / replays [backbone + part A of THIS step] beside [the rest of the PREVIOUS
        # step] - the post-backbone chain is ~45 small dependent launches that leave the chip idle - and returns the
        # previous step's results (None on the first call); `flush_steps()` finishes the pending step.  Off by default
        # because it changes what a call returns; bit-identical to the serial step (tests/test_head_gpu.py, also at the
        # bench size) since the ring barriers of the LDS-DMA kernels wait for the wave's own LDS reads
        self.overlap_steps = False
        self._ov_graphs = collections.OrderedDict()
        self._ov_pending = None
        self._ov_last = None
        self._pk_epoch = 0
        self._side_streams = {}
        self.side_stream = False
        # step graphs in flight: 2 = two captured instances of the step alternate on two streams, so the latency-bound
        # head of step i overlaps the backbone of step i+1 (results of a step are ordered on the caller's stream by
        # the next call or sync_steps()).  Measured on MI355X: no gain over depth 1 (2634 vs 2653 clips/s), so 1 is the default
        self.pipeline = 1
        self.side_fork_stage = 1      # fork after layer2 (0-based residual stage index; -1 = after the stem)
        # overlap_steps: where a step is cut into its two pipelined halves.  None: behind the backbone (the tail of step
        # n-1 runs beside the whole backbone of step n).  i (or (i, k): behind the first k blocks of stage i): behind
        # residual stage i - the first half is stem .. stage i
        # plus part A, the second half the remaining stages plus the tail, so that the HBM-bound early stages of step n
        # run beside the MFMA-bound late stages of step n-1 (tools/overlap_probe.py: 1834 -> 1581 us for the backbone)
        self.split_stage = None
        # pair attention + spatio_inferA/B as one kernel on [V | W_A V | W_B V] maps (see ops.pair_infer); off:
        # the reference's order of operations (pool, attend, then the two 1x1 convs as a GEMM)
        self.fuse_pair_infer = True
        # ... and that kernel on the matrix cores (ops.pair_infer_mfma: union-box pooling as a matrix product, scores /
        # softmax / combine in its epilogue) whenever the 2-D map is small enough (<= 16 x 16: TRACE's 14 x 14 maps)
        self.pair_mfma = True
        self.share_swapped_pairs = True  # (s,o) and (o,s) have one union box: pooled once for both
        # with side_stream: the union-box pooling of the pair kernel as its own launch on the second stream, only the
        # query-dependent combine after the tree-GRU.  "tail": pooled beside the post-backbone chain (temporal object
        # branch, tree-GRU, query GEMMs); True: beside the backbone (measured no faster than the fused launch: 2.869 vs
        # 2.852 ms/step); False: one fused launch after the tree-GRU
        self.split_pair_pool = "tail"
        # The networks around the hot path: `forward` calls them exactly where the reference does.
        # Conv_Body(data) -> blob_conv (:116, :702) = ResNet101_conv5_body with res5 dilated, Prd_RCNN2D(frame) -> feat_map
        # (:188-190, :1006-1007) = ResNet50_conv5_body_dilation2, both through the implicit-GEMM convs (resnet2d.py);
        # conv_bodies=False leaves them out (callers of forward_clips that bring blob_conv / feat_map themselves), and
        # any callable may be assigned instead.  RPN(blob_conv, im_info, roidb) -> rpn_ret (:733) stays pluggable.
        self.Conv_Body = resnet2d.ResNet101_conv5_body() if conv_bodies else None
        self.Prd_RCNN2D = resnet2d.ResNet50_conv5_body_dilation2() if conv_bodies else None
        self.RPN = None
        # a checkpoint loaded into a SUB-module (model.Box_Head.load_state_dict(..)) never reaches this class's
        # load_state_dict: every sub-module drops the packed weights (and the captured graphs, through the epoch in
        # their keys) from a post-hook of its own load
        for mod in self.modules():
            if mod is not self:
                mod.register_load_state_dict_post_hook(self._on_submodule_load)

    def _on_submodule_load(self, module, incompatible_keys):
        self._pk = {}
        self._pk_epoch = getattr(self, "_pk_epoch", 0) + 1

    # ------------------------------------------------------------------ packed-weight cache
    def _apply(self, fn, *a, **k):
        self._pk = {}
        self._pk_epoch = getattr(self, "_pk_epoch", 0) + 1
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._pk = {}
        self._pk_epoch = getattr(self, "_pk_epoch", 0) + 1
        return super().load_state_dict(*a, **k)

    def _tree_weights(self):
        tms = self.cluster_tree3d.tree_message_passing_list
        return self._cached("tree", lambda: tree_gru_weights(tms), [p for tm in tms for p in tm.parameters()])

    def _ffn12_w(self):
        tms = self.cluster_tree3d.tree_message_passing_list
        return self._cached("ffn12", lambda: ffn12_weights(tms), [p for tm in tms for p in tm.parameters()])

    # ------------------------------------------------------------------ host-side roi lists
    @staticmethod
    def _temporal_roi_list(boxes, T, frame0):
        """Temporal ROIAlign list of one clip in OBJECT-major output order (j, t).  The reference pairs
        frame t with the box of object (t*n + j) // T (cluster_tree3d.py:158-162 builds an object-major
        box list against frame-major ids, :181-182 views the result (T, n)); reproduced exactly."""
        n = boxes.shape[0]
        j, t = np.meshgrid(np.arange(n), np.arange(T), indexing="ij")
        src = (t * n + j) // T
        rois = np.empty((n, T, 5), np.float32)
        rois[:, :, 0] = frame0 + t
        rois[:, :, 1:] = boxes[src]
        return rois.reshape(-1, 5)

    # ------------------------------------------------------------------ forward
    _PLUGGABLE = ("Conv_Body", "Prd_RCNN2D", "RPN")

    def __setattr__(self, name, value):
        # the networks around the hot path may be any callable (a wrapped reference module, a stub in a test), not only
        # an nn.Module: a plain callable is kept as an ordinary attribute, outside the module tree / state dict
        if name in self._PLUGGABLE:
            if value is not None and not isinstance(value, nn.Module):
                self._modules.pop(name, None)
                object.__setattr__(self, name, value)
                return
            self.__dict__.pop(name, None)
        super().__setattr__(name, value)

    def forward_clips(self, all_frames, blob_conv, feat_map, det_rois, im_scale=1.0, f_scale=1.0, topk=0, prd_k=1,
                      det_labels=None, det_scores=None, pairs=None, pair_labels=None, lim_score=1.0):
        """Inference on B clips: the batched hot path behind `forward` (which keeps the reference's one-clip signature).
          all_frames (B,T,3,h,w) f32 GPU   the clip (model_builder_rel.py:688); or decoded uint8 frames (B,T,h,w,3), from
                                           which Prd_RCNN.pixel_means is subtracted inside the stem kernel
          blob_conv  (B,2048,H/16,W/16) f32 NCHW or (B,H/16,W/16,2048) bf16 — Conv_Body(data) (:702)
          feat_map   same layouts — Prd_RCNN2D(cur frame) (:1006-1007)
          det_rois   list of B numpy (n_b,5) [*,x1,y1,x2,y2] key-frame rois (:894)
          det_labels, det_scores   optional lists of B numpy (n_b,) arrays: labels (1-based) / scores that come
                     WITH the boxes - the `prd_category` records of pre-computed detections (:895-898) or the
                     detected-box branch's prepare_det_rois output (:912-915).  They then feed RelPN (on Action
                     Genome only persons, label <= 1, are subjects: relpn_heads.py:71-77) and the pair labels /
                     scores; without them labels / scores are the Box_Outs argmax (:900-904).
          pairs      optional list of B (sbj_inds, obj_inds) integer arrays into the clip's rois: the relation proposals
                     themselves - the gt-box branch (:772-888) pairs the de-duplicated gt boxes as annotated instead of
                     enumerating all ordered pairs (RelPN, :906).
          pair_labels optional list of B dicts(sbj_labels, obj_labels [1-based], sbj_scores, obj_scores), one entry per
                     pair: `use_gt_labels` (:844-858) - per-pair labels / scores that also index the frequency bias.
          lim_score  RelPN's argument of that name (relpn_heads.py:57, :94): the overlap filter is skipped below 0.015 -
                     the detected-box branch gets there when its score threshold had to be lowered (:911-935).
        Returns a list of B dicts with the reference's return keys (:1162-1182)."""
        if self.training:
            raise ValueError('You should call this function only on inference. '
                             'Set the network in inference mode by net.eval().')
        with torch.no_grad():
            dev = all_frames.device
            prep = self._host_prep(det_rois, self.Prd_RCNN.out_frames(all_frames.shape[1]), im_scale, f_scale,
                                   det_labels, det_scores, pairs, pair_labels, lim_score)
            if self.graph_step and self.overlap_steps and ops.IGEMM_TRACE is None:
                return self._replay_overlap(all_frames, blob_conv, feat_map, prep, topk, prd_k)
            if self._ov_pending is not None:
                raise RuntimeError("a step is pending in the overlapped pipeline: call flush_steps() first")
            if self.graph_step and ops.IGEMM_TRACE is None:
                res = self._replay_step(all_frames, blob_conv, feat_map, prep, topk, prd_k)
            else:
                dv = prep["upl"].commit(dev)
                res = self._device_forward(all_frames, self._nhwc(blob_conv), self._nhwc(feat_map), prep, dv, topk,
                                           prd_k, self.graph_backbone)
            return self._make_outs(prep, res, topk, prd_k)


    # ------------------------------------------------------------------ reference signature (one clip per call)
    @staticmethod
    def _unwrap(v):
        """The harness hands every input over as a one-element list for DataParallel's scatter
        (lib/core/test_rel.py:82-113); without that wrapper the elements arrive still wrapped."""
        while isinstance(v, (list, tuple)) and len(v) == 1:
            v = v[0]
        return v

    @staticmethod
    def _dataset_name(v, default):
        """`dataset_name` arrives serialized (utils/blob.py:165-176: the pickled string as a float32 array)."""
        if v is None:
            return default
        if isinstance(v, str):
            return v
        import io
        import pickle

        class _StrOnly(pickle.Unpickler):           # a pickled str needs no globals: refuse anything else
            def find_class(self, module, name):
                raise pickle.UnpicklingError("dataset_name: unexpected pickled object")
        raw = np.asarray(v.cpu() if torch.is_tensor(v) else v).astype(np.uint8).tobytes()
        out = _StrOnly(io.BytesIO(raw)).load()
        if not isinstance(out, str):
            raise ValueError("dataset_name must deserialize to a string")
        return out

    def roi_feature_transform(self, blobs_in, rpn_ret, blob_rois='rois', method='RoIAlign', resolution=7,
                              spatial_scale=1. / 16., sampling_ratio=0):
        """model_builder_rel.py:1342-1396 (FPN-aware dispatch or the single-level path)."""
        return roi_feature_transform(blobs_in, rpn_ret, blob_rois, method, resolution, spatial_scale, sampling_ratio)

    @staticmethod
    def gt_box_pairs(roidb, im_scale, use_gt_labels=False):
        """The gt-box branch's host arithmetic (model_builder_rel.py:776-810, 844-858): annotated (subject, object) box
        pairs de-duplicated by np.unique over the int32 8-tuples, the boxes themselves de-duplicated into det_rois,
        pair -> box indices, and (use_gt_labels) the labels / scores that come with the annotation.
        Returns dict(det_rois (n,5) f32 scaled, sbj_inds, obj_inds (P,), sbj_rois, obj_rois (P,5), rel_rois (P,5),
        pair_gt_index, and with use_gt_labels: sbj_labels, obj_labels, det_labels [0-based], *_scores)."""
        sbj_boxes = roidb['sbj_gt_boxes'].copy()
        obj_boxes = roidb['obj_gt_boxes'].copy()
        pair_gt_boxes = np.concatenate((sbj_boxes, obj_boxes), axis=1).astype(np.int32)
        pair_gt_boxes, pair_gt_index = np.unique(pair_gt_boxes, return_index=True, axis=0)
        sbj_boxes = pair_gt_boxes[:, :4].astype(np.float32)
        obj_boxes = pair_gt_boxes[:, 4:].astype(np.float32)
        all_boxes = np.concatenate((sbj_boxes, obj_boxes)).astype(np.int32)
        unique_all_boxes, gt_boxes_id = np.unique(all_boxes, return_inverse=True, axis=0)
        gt_boxes_id = np.asarray(gt_boxes_id).reshape(-1)           # numpy 2 returns the inverse shaped like the input
        sbj_id = gt_boxes_id[:sbj_boxes.shape[0]]
        obj_id = gt_boxes_id[sbj_boxes.shape[0]:]
        unique_all_boxes = unique_all_boxes.astype(np.float32)
        im_scale = np.float32(im_scale)
        zeros = lambda k: np.zeros((k, 1), np.float32)
        sbj_rois = np.hstack((zeros(len(sbj_boxes)), sbj_boxes * im_scale))
        obj_rois = np.hstack((zeros(len(obj_boxes)), obj_boxes * im_scale))
        all_rois = np.hstack((zeros(len(unique_all_boxes)), unique_all_boxes * im_scale))
        out = dict(det_rois=all_rois.astype(np.float32), sbj_inds=sbj_id, obj_inds=obj_id, sbj_rois=sbj_rois.astype(np.float32),
                   obj_rois=obj_rois.astype(np.float32), rel_rois=box_ops.rois_union_np(sbj_rois, obj_rois),
                   pair_gt_index=pair_gt_index)
        if use_gt_labels:
            sbj_labels = roidb['sbj_gt_classes'][pair_gt_index]     # start from 0
            obj_labels = roidb['obj_gt_classes'][pair_gt_index]
            all_labels = np.concatenate((sbj_labels, obj_labels))
            _, gt_labels_id = np.unique(gt_boxes_id, return_index=True)
            det_labels = all_labels[gt_labels_id]
            out.update(sbj_labels=sbj_labels, obj_labels=obj_labels, det_labels=det_labels,
                       sbj_scores=np.ones_like(sbj_labels, dtype=np.float32),
                       obj_scores=np.ones_like(obj_labels, dtype=np.float32),
                       det_scores=np.ones_like(det_labels, dtype=np.float32))
        return out

    def forward(self, data, all_frames, im_info, pre_processed_frames_rpn_ret=None, pre_processed_temporal_roi=None,
                do_vis=False, dataset_name=None, file_name=None, roidb=None, use_gt_labels=False, bf_cur_len=None,
                f_scale=None, **rpn_kwargs):
        """The reference's `Generalized_RCNN.forward` (model_builder_rel.py:648-662 -> `_forward` :664-1204), inference:
          data (1,3,H,W) f32 GPU key frame; all_frames (1,T,3,h,w) f32 GPU; im_info (1,3) [H,W,scale] (CPU);
          roidb dict with sbj_gt_boxes / obj_gt_boxes (/ *_gt_classes): the gt-box branch (:772-888);
          pre_processed_frames_rpn_ret {rois (n,5), labels_int32 [(n,)], (prd_category, score)}: pre-computed boxes
          (:704-731, :892-907); otherwise RPN + prepare_det_rois + RelPN (:733, :908-935).
        Returns the reference's dict (:1162-1182, :1195-1196); the entries the reference returns as numpy arrays and its
        harness reads through DataParallel's gather (`.data.cpu().numpy()`, lib/core/test_rel.py:120-125) are CPU
        tensors here, so `im_get_det_rels` works on the result without the DataParallel wrapper."""
        if self.training:
            raise NotImplementedError("trace_amd implements the inference path; call .eval() "
                                      "(the reference's helpers raise likewise outside inference, :91-104)")
        u = self._unwrap
        data, all_frames, im_info, roidb = u(data), u(all_frames), u(im_info), u(roidb)
        rpn_pre, use_gt_labels = u(pre_processed_frames_rpn_ret), bool(u(use_gt_labels))
        bf_cur_len, f_scale = u(bf_cur_len), u(f_scale)
        cfg = self.cfg
        dataset_name = self._dataset_name(u(dataset_name), cfg.dataset)
        if torch.is_tensor(bf_cur_len):
            bf_cur_len = int(bf_cur_len.item())
        if torch.is_tensor(f_scale):
            f_scale = float(f_scale.item())
        if bf_cur_len is None:
            bf_cur_len = all_frames.shape[1] // 2
        f_scale = 1.0 if f_scale is None else float(f_scale)
        if all_frames.shape[0] != 1:
            raise ValueError("forward() takes one clip per call as the reference does (relpn_heads.py:67); "
                             "forward_clips() is the batched entry")
        if self.Conv_Body is None or self.Prd_RCNN2D is None:
            raise RuntimeError("Generalized_RCNN.forward needs model.Conv_Body and model.Prd_RCNN2D (the 2-D conv bodies, "
                               "model_builder_rel.py:702,1006-1007): assign callables, or call forward_clips() with their outputs")
        im_info_np = np.asarray(im_info.detach().cpu().numpy() if torch.is_tensor(im_info) else im_info, np.float32)
        im_scale = float(im_info_np[0, 2])
        dev = all_frames.device
        with torch.no_grad():
            blob_conv = self.Conv_Body(data)                                                     # :702
            feat_map = self.Prd_RCNN2D(all_frames[:, int(bf_cur_len)])                           # :1006-1007
            pairs = pair_labels = det_labels = det_scores = None
            lim_score = cfg.TEST_SCORE_THRESH           # RelPN's lim_score (:906, :916): the threshold the detections passed
            bad_gt = roidb is None or (len(roidb['sbj_gt_boxes']) + len(roidb['obj_gt_boxes'])) <= 0   # :767-771
            if not bad_gt:
                # ---- gt-box branch (:772-888).  Box_Head / S_Head / O_Head pool sbj_rois / obj_rois / det_rois there;
                # sbj_rois[p] IS det_rois[sbj_inds[p]] bit for bit (same int32 box times the same f32 scale), so the
                # per-pair pooled rows and head outputs are gathers of the per-box ones: pooled once per unique box
                gt = self.gt_box_pairs(roidb, im_info_np[0, 2], use_gt_labels)
                det_rois = gt["det_rois"]
                pairs = [(gt["sbj_inds"], gt["obj_inds"])]
                if use_gt_labels:
                    det_labels, det_scores = [gt["det_labels"].astype(np.int64) + 1], [gt["det_scores"]]
                    pair_labels = [dict(sbj_labels=gt["sbj_labels"].astype(np.int64) + 1,
                                        obj_labels=gt["obj_labels"].astype(np.int64) + 1,
                                        sbj_scores=gt["sbj_scores"], obj_scores=gt["obj_scores"])]
            elif rpn_pre is not None and dataset_name.find('vidvrd') >= 0:
                # ---- pre-computed boxes (:704-731, :892-907); the reference scales the caller's array in place (:724)
                if dataset_name.find('vid') >= 0:
                    rpn_pre['rois'][:, 1:] *= float(im_info_np[0, -1])
                det_rois = np.asarray(rpn_pre['rois'], np.float32)
                if 'prd_category' in rpn_pre:
                    det_labels = [np.asarray(rpn_pre['prd_category']).astype(np.int64)]
                    det_scores = [np.amax(np.asarray(rpn_pre['score'], np.float32)[:, 1:], axis=1)]
            else:
                # ---- detected boxes (:733, :908-935): RPN -> Box_Head / Box_Outs on every proposal -> per-class NMS
                if rpn_pre is not None:
                    # pre-processed rois on another dataset than VidVRD (:704-731 eval branch): they stand in for the RPN's
                    # proposals and then go through the detection heads + NMS like proposals (:908-935)
                    if dataset_name.find('vid') >= 0:
                        rpn_pre['rois'][:, 1:] *= float(im_info_np[0, -1])
                    rpn_ret = {'rois': np.asarray(rpn_pre['rois'], np.float32)}
                elif self.RPN is None:
                    raise RuntimeError("Generalized_RCNN.forward: no gt boxes and no pre-computed boxes were given, and "
                                       "model.RPN is not set (RPN + generate_proposals: model_builder_rel.py:733)")
                else:
                    rpn_ret = self.RPN(blob_conv, im_info, roidb)
                box_feat = self.Box_Head(blob_conv, rpn_ret, use_relu=True)                      # :742
                cls_score, bbox_pred = self.Box_Outs(box_feat)                                   # :743
                rois_d = torch.from_numpy(np.asarray(rpn_ret['rois'], np.float32)).to(dev)
                score_thresh, iter_num = cfg.TEST_SCORE_THRESH, 0
                while score_thresh >= -1e-05:                                                    # :911-935
                    d_rois, d_lab, d_sc, _ = box_ops.prepare_det_rois(
                        rois_d, cls_score, bbox_pred, im_info_np, score_thresh, cfg.TEST_NMS, cfg.DETECTIONS_PER_IM, 1)
                    # the reference leaves the loop when RelPN returns at least one pair (:916-920).  RelPN pairs a single
                    # detection with itself (relpn_heads.py:79-84 removes self pairs only when there are two or more) and
                    # its overlap filter never empties the list (:99), so that is exactly "at least one detection"
                    if d_rois.shape[0] > 0:
                        break
                    if score_thresh >= 0.02 - 1e-04:
                        score_thresh -= 0.01
                    elif iter_num <= 50:
                        score_thresh /= 2.0
                        iter_num += 1
                    else:
                        score_thresh = -2e-06
                lim_score = score_thresh
                det_rois = d_rois.cpu().numpy().astype(np.float32)
                det_labels = [d_lab.cpu().numpy().astype(np.int64)]
                det_scores = [d_sc.cpu().numpy().astype(np.float32)]
            outs = self.forward_clips(all_frames, blob_conv, feat_map, [det_rois], im_scale=im_scale, f_scale=f_scale,
                                      det_labels=det_labels, det_scores=det_scores, pairs=pairs, pair_labels=pair_labels,
                                      lim_score=lim_score)
            if outs is None:                        # overlap_steps returns the previous step: one clip per call wants its own
                outs = self.flush_steps()
            o = outs[0]
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
            ret = {}
            ret['sbj_rois'], ret['obj_rois'] = t(o['sbj_rois']), t(o['obj_rois'])
            ret['sbj_labels'] = o['sbj_labels'].to(torch.int32).cpu()                            # 1-based (:873-874)
            ret['obj_labels'] = o['obj_labels'].to(torch.int32).cpu()
            ret['sbj_scores'], ret['obj_scores'] = o['sbj_scores'].float().cpu(), o['obj_scores'].float().cpu()
            ret['det_rois'] = t(o['det_rois'])
            ret['det_labels'] = o['det_labels'].to(torch.int32).cpu()
            ret['msp_det_scores'] = o['msp_det_scores']
            ret['sbj_inds'], ret['obj_inds'] = t(np.asarray(o['sbj_inds']).reshape(-1)), t(np.asarray(o['obj_inds']).reshape(-1))
            ret['prd_scores'] = o['prd_scores']
            if cfg.USE_FREQ_BIAS:                                                                # :1064, :1085-1086, :1178
                lab = torch.stack((o['sbj_labels'].long() - 1, o['obj_labels'].long() - 1), 1)
                bias = self.freq_bias.rel_index_with_labels(lab)
                ret['prd_scores_bias'] = torch.sigmoid(bias) if cfg.MULTI_RELATION else F.softmax(bias, dim=1)
            ret['prd_ttl_scores'] = o['prd_scores']                                              # ADD_SCORES_ALL (:1181-1182)
            if do_vis:
                ret['blob_conv'], ret['feat_map'] = blob_conv, feat_map
                ret['temporal_blob_conv_prd'] = self._debug["tmaps"].permute(0, 1, 4, 2, 3).float()
            ret['pre_processed_frames_rpn_ret'] = None                                           # :1195-1196
            ret['pre_processed_temporal_roi'] = None
            return ret

    # ------------------------------------------------------------------ whole-step HIP graph
    def _replay_step(self, all_frames, blob_conv, feat_map, prep, topk, prd_k):
        """One captured graph per step signature (clip shape, objects / tree regions per clip): every launch
        of the step - backbone, ROIAligns, GEMMs, attention, tree-GRU, ranking - is replayed with one call.
        Per step only the packed host arrays (one pinned copy) and, unless the caller already writes
        into the graph's buffers (`step_buffers`), the clip and the two 2-D maps are copied in.  The
        returned tensors are the graph's output buffers: they are overwritten by the next replay."""
        dev = all_frames.device
        key = (tuple(all_frames.shape), all_frames.dtype, tuple(blob_conv.shape), blob_conv.dtype, tuple(feat_map.shape),
               feat_map.dtype, prep["sig"], topk, prd_k, dev.index, _PACK_EPOCH[0], self._pk_epoch,
               self.side_stream, self.split_pair_pool, self.fuse_pair_infer, self.share_swapped_pairs, self.side_fork_stage,
               self.pair_mfma, self.split_stage)
        depth = max(1, int(self.pipeline))
        ents = self._step_graphs.get(key)
        if ents is None:
            self._evict(self._step_graphs, dev)
            ents = self._step_graphs[key] = dict(inst=[None] * depth, calls=0)
        else:
            self._step_graphs.move_to_end(key)
        if len(ents["inst"]) != depth:
            raise RuntimeError("pipeline depth changed under captured step graphs: clear model._step_graphs first")
        k = ents["calls"] % depth
        ent = ents["inst"][k]
        cur = torch.cuda.current_stream(dev)
        if ent is None:
            frames_s = all_frames.detach().clone(memory_format=torch.contiguous_format)
            blob_s, fmap_s = self._nhwc(blob_conv).clone(), self._nhwc(feat_map).clone()
            nb = (max(prep["upl"].nbytes, 256) + 255) // 256 * 256
            dv_buf = torch.empty((nb,), dtype=torch.uint8, device=dev)
            # the step's first graph node reads the packed host arrays from a two-slot pinned buffer
            # (ops.upload_slots): no copy-engine hop between two replays, and the host fills slot k+1 while
            # step k still runs
            pinned = torch.zeros((2 * nb,), dtype=torch.uint8, pin_memory=True)
            counter = torch.zeros((2,), dtype=torch.int32, device=dev)   # [launch number, arrival count]
            torch.cuda.synchronize(dev)              # earlier instances may be running on their own streams
            with ops.workspace_base(2 * k):          # concurrent instances keep their own split-K scratch
                dv = prep["upl"].commit(dev, into=dv_buf[:max(prep["upl"].nbytes, 256)])
                side = torch.cuda.Stream(device=dev)
                side.wait_stream(cur)
                with torch.cuda.stream(side):        # packs weights, sizes workspaces, warms the allocator
                    self._device_forward(frames_s, blob_s, fmap_s, prep, dv, topk, prd_k, False)
                cur.wait_stream(side)
                dv = prep["upl"].views(dv_buf)
                g = torch.cuda.CUDAGraph()
                with ops.capture(g):
                    ops.upload_slots(pinned, nb, counter, dv_buf, nb)
                    res = self._device_forward(frames_s, blob_s, fmap_s, prep, dv, topk, prd_k, False)
            ent = dict(graph=g, frames=frames_s, blob=blob_s, fmap=fmap_s, dv=dv_buf, res=res, pinned=pinned,
                       pinned_np=pinned.numpy(), counter=counter, slot_bytes=nb, seq=0, events=[None, None],
                       nbytes=prep["upl"].nbytes, stream=torch.cuda.Stream(device=dev) if depth > 1 else None,
                       done=None, owner_key=key)
            ents["inst"][k] = ent
        if prep["upl"].nbytes != ent["nbytes"]:
            raise RuntimeError("upload layout changed under a captured step")
        slot = ent["seq"] & 1
        ev = ent["events"][slot]
        if ev is not None:
            ev.synchronize()                         # the replay that read this slot two steps ago has finished
        prep["upl"].fill(ent["pinned_np"][slot * ent["slot_bytes"]:(slot + 1) * ent["slot_bytes"]])
        run = ent["stream"] if ent["stream"] is not None else cur
        copied = False
        if run is not cur and ent["done"] is not None:
            cur.wait_event(ent["done"])              # input copies below must not overtake this instance's last replay
        if all_frames.data_ptr() != ent["frames"].data_ptr():
            ent["frames"].copy_(all_frames)
            copied = True
        if blob_conv.data_ptr() != ent["blob"].data_ptr():
            ent["blob"].copy_(self._nhwc(blob_conv))
            copied = True
        if feat_map.data_ptr() != ent["fmap"].data_ptr():
            ent["fmap"].copy_(self._nhwc(feat_map))
            copied = True
        if run is not cur and copied:
            run.wait_stream(cur)
        with torch.cuda.stream(run):
            ent["graph"].replay()
            ev = ent["events"][slot]
            if ev is None:
                ev = ent["events"][slot] = torch.cuda.Event()
            ev.record()
        ent["done"] = ev
        # pipeline depth > 1: the caller's stream is ordered behind the PREVIOUS step only, so this step overlaps
        # the next one; its results are valid on the caller's stream after the next call or `sync_steps()`
        prev = self._last_step_entry
        if run is not cur and prev is not None and prev is not ent and prev["done"] is not None:
            cur.wait_event(prev["done"])
        ent["seq"] += 1
        ents["calls"] += 1
        self._last_step_entry = ent
        return ent["res"]

    # ------------------------------------------------------------------ software-pipelined step graphs
    def _replay_overlap(self, all_frames, blob_conv, feat_map, prep, topk, prd_k):
        """`overlap_steps`: two captured graphs per step signature alternate.  Graph w uploads the packed host arrays
        of step n into buffer w, runs the backbone + part A of step n (outputs = hand-off set w) and, on a forked
        branch with its own split-K scratch, the rest of step n-1 from hand-off set 1-w.  Graph 0 is captured first,
        so its tail reads a static copy that graph 1 fills after its part A; graph 1's tail reads graph 0's outputs
        in place.  Returns the per-clip dicts of step n-1 (None when nothing was pending)."""
        dev = all_frames.device
        key = (tuple(all_frames.shape), all_frames.dtype, tuple(blob_conv.shape), blob_conv.dtype, tuple(feat_map.shape),
               feat_map.dtype, prep["sig"], topk, prd_k, dev.index, _PACK_EPOCH[0], self._pk_epoch,
               self.side_stream, self.split_pair_pool, self.fuse_pair_infer, self.share_swapped_pairs, self.side_fork_stage,
               self.pair_mfma, self.split_stage)
        flushed = None
        if self._ov_pending is not None and self._ov_pending["key"] != key:
            flushed = self.flush_steps()          # another step signature: no graph holds that pair of steps
        ent = self._ov_graphs.get(key)
        cur = torch.cuda.current_stream(dev)
        if ent is not None:
            self._ov_graphs.move_to_end(key)
        if ent is None:
            self._evict(self._ov_graphs, dev)
            frames_s = all_frames.detach().clone(memory_format=torch.contiguous_format)
            blob_s, fmap_s = self._nhwc(blob_conv).clone(), self._nhwc(feat_map).clone()
            nb = (max(prep["upl"].nbytes, 256) + 255) // 256 * 256
            nreal = max(prep["upl"].nbytes, 256)
            var = []
            for w in range(2):
                var.append(dict(dv=torch.empty((nb,), dtype=torch.uint8, device=dev),
                                pinned=torch.zeros((2 * nb,), dtype=torch.uint8, pin_memory=True),
                                counter=torch.zeros((2,), dtype=torch.int32, device=dev), seq=0, events=[None, None]))
                var[w]["pinned_np"] = var[w]["pinned"].numpy()
            torch.cuda.synchronize(dev)
            prep["upl"].commit(dev, into=var[0]["dv"][:nreal])
            var[1]["dv"].copy_(var[0]["dv"])
            dvv = [prep["upl"].views(var[0]["dv"]), prep["upl"].views(var[1]["dv"])]
            warm = torch.cuda.Stream(device=dev)
            warm.wait_stream(cur)
            with torch.cuda.stream(warm):        # packs weights, sizes workspaces, warms the allocator (both scratch bases)
                static = self._device_forward(frames_s, blob_s, fmap_s, prep, dvv[1], topk, prd_k, False, phase="a")
                with ops.workspace_base(2):
                    self._device_forward(None, blob_s, fmap_s, prep, dvv[1], topk, prd_k, False, phase="b", hand=static)
            cur.wait_stream(warm)
            torch.cuda.synchronize(dev)
            tail_stream = torch.cuda.Stream(device=dev)
            graphs, hands, results = [], [None, static], []
            for w in range(2):
                g = torch.cuda.CUDAGraph()
                with ops.capture(g):
                    main = torch.cuda.current_stream(dev)
                    # the upload is the graph's one root node: the tail (which needs nothing of this step) forks behind
                    # it - forking from the still empty capture stream and forking again before any launch on the
                    # branch crashed hipStreamEndCapture
                    ops.upload_slots(var[w]["pinned"], nb, var[w]["counter"], var[w]["dv"], nb)
                    tail_stream.wait_stream(main)
                    with torch.cuda.stream(tail_stream), ops.workspace_base(2):
                        res = self._device_forward(None, blob_s, fmap_s, prep, dvv[1 - w], topk, prd_k, False, phase="b",
                                                   hand=hands[1 - w])
                    out = self._device_forward(frames_s, blob_s, fmap_s, prep, dvv[w], topk, prd_k, False, phase="a")
                    if w == 0:
                        hands[0] = out
                    else:
                        for name, dst in static.items():               # graph 0 was captured against the static set
                            if torch.is_tensor(dst):
                                dst.copy_(out[name])
                    main.wait_stream(tail_stream)
                graphs.append(g)
                results.append(res)
            # the first half alone, for a call that finds the pipeline empty (the first call, the call behind a flush):
            # the full graph would run its tail branch on a stale hand-off set beside it - harmless, but 0.8 ms of
            # chip time per pipeline fill.  Same hand-off targets as the full graphs (graph 0's own outputs / the
            # static set), filled by copies
            agraphs = []
            for w in range(2):
                g = torch.cuda.CUDAGraph()
                with ops.capture(g):
                    ops.upload_slots(var[w]["pinned"], nb, var[w]["counter"], var[w]["dv"], nb)
                    out = self._device_forward(frames_s, blob_s, fmap_s, prep, dvv[w], topk, prd_k, False, phase="a")
                    for name, dst in hands[w].items():
                        if torch.is_tensor(dst):
                            dst.copy_(out[name])
                agraphs.append(g)
            # the tail alone, for flush_steps(): one graph per hand-off set
            fgraphs, fres = [], []
            for w in range(2):
                g = torch.cuda.CUDAGraph()
                with ops.capture(g):
                    fres.append(self._device_forward(None, blob_s, fmap_s, prep, dvv[w], topk, prd_k, False, phase="b",
                                                     hand=hands[w]))
                fgraphs.append(g)
            ent = self._ov_graphs[key] = dict(fgraphs=fgraphs, fres=fres, agraphs=agraphs,
                                              graphs=graphs, hands=hands, res=results, var=var, dvv=dvv, frames=frames_s,
                                              blob=blob_s, fmap=fmap_s, nbytes=prep["upl"].nbytes, slot_bytes=nb, n=0,
                                              keep=(static, tail_stream))
        if prep["upl"].nbytes != ent["nbytes"]:
            raise RuntimeError("upload layout changed under a captured step")
        w = ent["n"] & 1
        v = ent["var"][w]
        slot = v["seq"] & 1
        ev = v["events"][slot]
        if ev is not None:
            ev.synchronize()                         # the replay that read this pinned slot has finished
        prep["upl"].fill(v["pinned_np"][slot * ent["slot_bytes"]:(slot + 1) * ent["slot_bytes"]])
        if all_frames.data_ptr() != ent["frames"].data_ptr():
            ent["frames"].copy_(all_frames)
        if blob_conv.data_ptr() != ent["blob"].data_ptr():
            ent["blob"].copy_(self._nhwc(blob_conv))
        if feat_map.data_ptr() != ent["fmap"].data_ptr():
            ent["fmap"].copy_(self._nhwc(feat_map))
        (ent["graphs"] if self._ov_pending is not None else ent["agraphs"])[w].replay()
        if ev is None:
            ev = v["events"][slot] = torch.cuda.Event()
        ev.record()
        v["seq"] += 1
        ent["n"] += 1
        self._ov_last = ent
        pend = self._ov_pending
        outs = flushed
        if pend is not None:
            outs = self._make_outs(pend["prep"], ent["res"][w], topk, prd_k)
        self._ov_pending = dict(key=key, ent=ent, w=w, prep=prep, topk=topk, prd_k=prd_k)
        return outs

    def _evict(self, cache, dev):
        """Make room for one more captured step signature: drop the least recently used entries beyond
        `max_step_graphs` - 1.  A graph is only destroyed on an idle device: nothing of it may still be in flight when
        hipGraphExecDestroy runs, and the memory of its private pool (which tensors handed to the caller may still view)
        goes back to the allocator only when the last such tensor dies."""
        limit = max(1, int(self.max_step_graphs))
        if len(cache) < limit:
            return
        if cache is self._ov_graphs and self._ov_pending is not None:
            raise RuntimeError("a step is pending in the overlapped pipeline: call flush_steps() first")   # callers flush on a signature change
        torch.cuda.synchronize(dev)
        while len(cache) >= limit:
            _, ent = cache.popitem(last=False)
            if cache is self._step_graphs:
                if any(self._last_step_entry is e for e in ent["inst"]):
                    self._last_step_entry = None
            elif self._ov_last is ent:
                self._ov_last = None
            del ent
        gc.collect()               # graph objects die here, on the idle device - never later inside a capture (ops.capture)
        torch.cuda.synchronize(dev)

    def drop_graphs(self):
        """Destroy every captured step graph (finishing a pending overlapped step first, whose results are returned).
        Tensors of earlier results stay valid; their memory is released when they die."""
        outs = self.flush_steps()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        self._step_graphs.clear()
        self._ov_graphs.clear()
        self._last_step_entry = None
        self._ov_last = None
        gc.collect()               # as in _evict
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        return outs

    def flush_steps(self):
        """Finish the step still pending in the overlapped pipeline (a captured graph of the tail alone, on the current
        stream) and return its per-clip dicts; None when nothing is pending."""
        pend = self._ov_pending
        if pend is None:
            return None
        ent, w = pend["ent"], pend["w"]
        ent["fgraphs"][w].replay()
        self._ov_pending = None
        return self._make_outs(pend["prep"], ent["fres"][w], pend["topk"], pend["prd_k"])

    def sync_steps(self):
        """Order the current stream behind every step still running on a pipeline stream (pipeline depth > 1)."""
        for ents in self._step_graphs.values():
            for ent in ents["inst"]:
                if ent is not None and ent["done"] is not None and ent["stream"] is not None:
                    torch.cuda.current_stream(ent["frames"].device).wait_event(ent["done"])

    def step_buffers(self):
        """(clip, blob_conv, feat_map) input buffers of the step-graph instance the NEXT call with the last step
        signature will replay: a loader that fills them in place (channels-last bf16 maps) saves the per-step
        copies.  With pipeline depth > 1 the instances alternate, so ask before every step."""
        if self.overlap_steps and self._ov_last is not None:
            e = self._ov_last
            return e["frames"], e["blob"], e["fmap"]
        e = self._last_step_entry
        ents = self._step_graphs[e["owner_key"]]
        nxt = ents["inst"][ents["calls"] % len(ents["inst"])]
        if nxt is None:
            nxt = e
        return nxt["frames"], nxt["blob"], nxt["fmap"]

    @staticmethod
    def _nhwc(x):
        if x.dtype == torch.bfloat16 and x.shape[-1] >= x.shape[1]:
            return x.contiguous()
        return x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16)

    def _host_prep(self, det_rois, Tn, im_scale, f_scale, det_labels=None, det_scores=None, pairs_in=None,
                   pair_labels=None, lim_score=1.0):
        """Host side of a step: roi lists, box trees, pair lists (numpy, as in the reference), packed for
        one upload.  `sig` identifies every launch shape of the step."""
        cfg = self.cfg
        B = len(det_rois)
        n_list = [int(r.shape[0]) for r in det_rois]
        off = np.concatenate([[0], np.cumsum(n_list)]).astype(np.int64)
        Ntot = int(off[-1])
        det_all = np.concatenate([np.concatenate([np.full((r.shape[0], 1), b, np.float32),
                                                  np.asarray(r[:, 1:5], np.float32)], 1)
                                  for b, r in enumerate(det_rois)], 0)
        scale3d = float(1.0 * f_scale * (1.0 / im_scale))                        # :1766-1770
        if (det_labels is None) != (det_scores is None):
            raise ValueError("det_labels and det_scores come together")
        person_sbj = det_labels is not None and cfg.dataset.find("ag") >= 0
        if pair_labels is not None and pairs_in is None:
            raise ValueError("pair_labels come with explicit pairs")
        if B > 1 and len(set(n_list)) == 1 and not cfg.USE_OVLP_FILTER and det_labels is None and pairs_in is None:
            return self._host_prep_same_n(det_all, B, n_list[0], Tn, scale3d, off)
        trees, t_obj, t_reg, reg_rois, pairs = [], [], [], [], []
        m_list = []
        for b in range(B):
            roi = det_all[off[b]:off[b + 1]].copy()
            roi[:, 1:5] *= scale3d
            boxes = roi[:, 1:5].astype(np.float32)
            tree = ct.build_tree(boxes, cfg.MSP_MAX_TREE_HEIGHT, cfg.CLUSTER_NUM_PARTITION)
            trees.append(tree)
            n = n_list[b]
            m = tree["parent"].shape[0] - n
            m_list.append(m)
            t_obj.append(self._temporal_roi_list(boxes, Tn, b * Tn))
            # tracklets are static tubes (model_builder_rel.py:1441-1444), so the per-frame hull of a region
            # (cluster_tree3d.py:368-391: min / max over its sons, exact) is the region's tree box on every
            # frame; the list keeps the reference's object-major-vs-frame-major pairing (same quirk as objects)
            t_reg.append(self._temporal_roi_list(tree["boxes"][n:].astype(np.float32), Tn, b * Tn))
            reg_rois.append(np.concatenate([np.full((m, 1), b, np.float32), tree["boxes"][n:].astype(np.float32)], 1))
            if pairs_in is not None:                                             # gt-box branch: pairs as annotated (:809-810)
                s = np.asarray(pairs_in[b][0], np.int64).reshape(-1)
                o = np.asarray(pairs_in[b][1], np.int64).reshape(-1)
                if len(s) != len(o) or len(s) == 0 or s.min() < 0 or o.min() < 0 or max(s.max(), o.max()) >= n:
                    raise ValueError("pairs[%d]: need equally long, non-empty index lists into the clip's %d rois" % (b, n))
            elif person_sbj or cfg.USE_OVLP_FILTER:
                # RelPN (relpn_heads.py:71-105): person subjects on Action Genome, then the overlap filter on the
                # boxes in original image scale
                s, o = box_ops.relpn_pair_indices(n, det_labels[b] if person_sbj else None, cfg.dataset)
                if cfg.USE_OVLP_FILTER and lim_score >= 0.015:
                    bx = (det_all[off[b]:off[b + 1], 1:5] / np.float32(im_scale)).astype(np.float32)
                    keep = box_ops.ovlp_filter(bx[s], bx[o])
                    s, o = s[keep], o[keep]
            elif n > 1:
                s = np.repeat(np.arange(n), n - 1)
                o = np.tile(np.arange(n - 1), n)
                o = o + (o >= s)
            else:
                s = o = np.zeros(1, np.int64)
            union = np.stack([np.full(len(s), b, np.float32), np.minimum(roi[s, 1], roi[o, 1]),
                              np.minimum(roi[s, 2], roi[o, 2]), np.maximum(roi[s, 3], roi[o, 3]),
                              np.maximum(roi[s, 4], roi[o, 4])], 1).astype(np.float32)
            pairs.append((s, o, ct.lca(tree, s, o) - n, union))
        moff = np.concatenate([[0], np.cumsum(m_list)]).astype(np.int64)
        Mtot = int(moff[-1])
        node_obj = np.concatenate([np.arange(n_list[b]) + off[b] + moff[b] for b in range(B)])
        node_reg = np.concatenate([np.arange(m_list[b]) + off[b + 1] + moff[b] for b in range(B)])
        poff = np.concatenate([[0], np.cumsum([len(p[0]) for p in pairs])]).astype(np.int64)
        upl = _Upload()
        forest_np = ct.forest_arrays(trees)
        forest_keys = list(forest_np)
        forest_ix = [upl.add(forest_np[k]) for k in forest_keys]
        ix = dict(det=upl.add(det_all), node_obj=upl.add(node_obj), node_reg=upl.add(node_reg),
                  X=upl.add(np.concatenate([p[0] + off[b] for b, p in enumerate(pairs)]).astype(np.int64)),
                  Y=upl.add(np.concatenate([p[1] + off[b] for b, p in enumerate(pairs)]).astype(np.int64)),
                  LCA=upl.add(np.concatenate([p[2] + moff[b] for b, p in enumerate(pairs)]).astype(np.int64)),
                  X32=upl.add(np.concatenate([p[0] + off[b] for b, p in enumerate(pairs)]).astype(np.int32)),
                  Y32=upl.add(np.concatenate([p[1] + off[b] for b, p in enumerate(pairs)]).astype(np.int32)),
                  PART32=upl.add(swapped_pair_index(np.concatenate([p[0] + off[b] for b, p in enumerate(pairs)]),
                                                    np.concatenate([p[1] + off[b] for b, p in enumerate(pairs)]))),
                  union=upl.add(np.concatenate([p[3] for p in pairs], 0)),
                  t_obj=upl.add(np.concatenate(t_obj, 0)), t_reg=upl.add(np.concatenate(t_reg, 0)),
                  reg=upl.add(np.concatenate(reg_rois, 0)),
                  pclip=upl.add(np.repeat(np.arange(B), np.diff(poff)).astype(np.int64)),
                  poff=upl.add(poff.astype(np.int64)))
        ix["PWORK"] = upl.add(ops.pair_work_records(upl.items[ix["X32"]][1], upl.items[ix["Y32"]][1], upl.items[ix["union"]][1],
                                                    upl.items[ix["PART32"]][1] if self.share_swapped_pairs else None))
        n_work = int(upl.items[ix["PWORK"]][1].shape[0])
        if det_labels is not None:
            lab = np.concatenate([np.asarray(l).reshape(-1) for l in det_labels]).astype(np.int64)
            sc = np.concatenate([np.asarray(v).reshape(-1) for v in det_scores]).astype(np.float32)
            if lab.shape[0] != Ntot or sc.shape[0] != Ntot:
                raise ValueError("det_labels / det_scores must hold one entry per roi")
            ix["lab"], ix["sc"] = upl.add(lab), upl.add(sc)
        if pair_labels is not None:
            Ptot = int(poff[-1])
            cat = lambda k, dt: np.concatenate([np.asarray(d[k]).reshape(-1) for d in pair_labels]).astype(dt)
            pl = dict(plab_s=cat("sbj_labels", np.int64), plab_o=cat("obj_labels", np.int64),
                      psc_s=cat("sbj_scores", np.float32), psc_o=cat("obj_scores", np.float32))
            if any(v.shape[0] != Ptot for v in pl.values()):
                raise ValueError("pair_labels must hold one entry per pair")
            for k, v in pl.items():
                ix[k] = upl.add(v)
        return dict(B=B, Tn=Tn, n_list=n_list, m_list=m_list, off=off, moff=moff, poff=poff, Ntot=Ntot, Mtot=Mtot,
                    det_all=det_all, pairs=pairs, upl=upl, ix=ix, forest_keys=forest_keys, forest_ix=forest_ix,
                    sig=(Tn, tuple(n_list), tuple(m_list), tuple(len(p[0]) for p in pairs), det_labels is not None,
                         pair_labels is not None, n_work),
                    max_tree_nodes=max(n + m for n, m in zip(n_list, m_list)))

    def _host_prep_same_n(self, det_all, B, n, Tn, scale3d, off):
        """`_host_prep` when every clip has n boxes: each numpy call runs once over a leading clip axis
        (cluster_tree.build_trees_same_n) instead of once per clip; the packed arrays are byte-identical."""
        cfg = self.cfg
        roi = det_all.reshape(B, n, 5).copy()
        roi[:, :, 1:5] *= scale3d
        boxes = roi[:, :, 1:5].astype(np.float32)
        bt = ct.build_trees_same_n(boxes, cfg.MSP_MAX_TREE_HEIGHT, cfg.CLUSTER_NUM_PARTITION)
        m = bt["parent"].shape[1] - n
        rboxes = bt["boxes"][:, n:].astype(np.float32)                          # (B,m,4)
        frame0 = (np.arange(B) * Tn).astype(np.float32)

        def temporal(bx):                                                        # bx (B,k,4) -> (B*k*Tn, 5)
            k = bx.shape[1]
            j, t = np.meshgrid(np.arange(k), np.arange(Tn), indexing="ij")
            src = (t * k + j) // Tn
            r = np.empty((B, k, Tn, 5), np.float32)
            r[:, :, :, 0] = frame0[:, None, None] + t[None].astype(np.float32)
            r[:, :, :, 1:] = bx[:, src]
            return r.reshape(-1, 5)

        t_obj, t_reg = temporal(boxes), temporal(rboxes)
        reg = np.empty((B, m, 5), np.float32)
        reg[:, :, 0] = np.arange(B, dtype=np.float32)[:, None]
        reg[:, :, 1:] = rboxes
        if n > 1:
            sidx = np.repeat(np.arange(n), n - 1)
            oidx = np.tile(np.arange(n - 1), n)
            oidx = oidx + (oidx >= sidx)
        else:
            sidx = oidx = np.zeros(1, np.int64)
        P = len(sidx)
        rs, ro = roi[:, sidx], roi[:, oidx]
        union = np.empty((B, P, 5), np.float32)
        union[:, :, 0] = np.arange(B, dtype=np.float32)[:, None]
        union[:, :, 1:3] = np.minimum(rs[:, :, 1:3], ro[:, :, 1:3])
        union[:, :, 3:5] = np.maximum(rs[:, :, 3:5], ro[:, :, 3:5])
        lca = ct.lca_same_n(bt, sidx, oidx) - n                                 # (B,P) region index inside the clip
        m_list = [m] * B
        moff = (np.arange(B + 1) * m).astype(np.int64)
        poff = (np.arange(B + 1) * P).astype(np.int64)
        node_obj = (np.arange(n)[None, :] + (off[:B] + moff[:B])[:, None]).reshape(-1)
        node_reg = (np.arange(m)[None, :] + (off[1:] + moff[:B])[:, None]).reshape(-1)
        X = (sidx[None, :] + off[:B, None]).reshape(-1)
        Y = (oidx[None, :] + off[:B, None]).reshape(-1)
        upl = _Upload()
        forest_np = ct.forest_arrays_same_n(bt)
        forest_keys = list(forest_np)
        forest_ix = [upl.add(forest_np[k]) for k in forest_keys]
        ix = dict(det=upl.add(det_all), node_obj=upl.add(node_obj), node_reg=upl.add(node_reg),
                  X=upl.add(X.astype(np.int64)), Y=upl.add(Y.astype(np.int64)),
                  LCA=upl.add((lca + moff[:B, None]).reshape(-1).astype(np.int64)),
                  X32=upl.add(X.astype(np.int32)), Y32=upl.add(Y.astype(np.int32)),
                  PART32=upl.add(swapped_pair_index(X, Y)),
                  union=upl.add(union.reshape(-1, 5)), t_obj=upl.add(t_obj), t_reg=upl.add(t_reg),
                  reg=upl.add(reg.reshape(-1, 5)),
                  pclip=upl.add(np.repeat(np.arange(B), P).astype(np.int64)), poff=upl.add(poff))
        ix["PWORK"] = upl.add(ops.pair_work_records(upl.items[ix["X32"]][1], upl.items[ix["Y32"]][1], upl.items[ix["union"]][1],
                                                    upl.items[ix["PART32"]][1] if self.share_swapped_pairs else None))
        n_work = int(upl.items[ix["PWORK"]][1].shape[0])
        pairs = [(sidx, oidx, None, None)] * B
        n_list = [n] * B
        return dict(B=B, Tn=Tn, n_list=n_list, m_list=m_list, off=off, moff=moff, poff=poff, Ntot=B * n, Mtot=B * m,
                    det_all=det_all, pairs=pairs, upl=upl, ix=ix, forest_keys=forest_keys, forest_ix=forest_ix,
                    sig=(Tn, tuple(n_list), tuple(m_list), (P,) * B, False, False, n_work), max_tree_nodes=n + m)

    def _device_forward(self, all_frames, blob, fmap, prep, dv, topk, prd_k, graph_backbone, phase="all", hand=None):
        """Device side of a step: launches only (no host synchronisation, no host->device copies), so the
        whole of it can be captured into a HIP graph.  Returns flat tensors over all clips.
        phase "a": stop after the backbone and part A and return what the rest of the step reads (`hand`: the
        temporal maps + part A's outputs); phase "b": the rest of the step from such a `hand` (`overlap_steps`)."""
        cfg = self.cfg
        dev = blob.device
        B = blob.shape[0]
        ps = self.pool_size
        D, low = cfg.MLP_HEAD_DIM, cfg.LOW_FEAT_DIM
        C = cfg.CONV_BODY_DIM
        ct3 = self.cluster_tree3d
        ix, Ntot, Mtot, poff = prep["ix"], prep["Ntot"], prep["Mtot"], prep["poff"]
        forest = {k: dv[i] for k, i in zip(prep["forest_keys"], prep["forest_ix"])}
        det_all_d, node_obj_d, node_reg_d = dv[ix["det"]], dv[ix["node_obj"]], dv[ix["node_reg"]]
        X, Y, LCA, union_d = dv[ix["X"]], dv[ix["Y"]], dv[ix["LCA"]], dv[ix["union"]]
        E = cfg.OBJ_EMBED_DIM

        # ---- everything that does not read the temporal maps runs as `part_a`; with `side_stream` on a second
        # stream beside the backbone (measured inside the captured step graph: 3.00 -> 2.93 ms per step, the same
        # for every fork point between the stem and layer3; an early version of the head lost 10 % this way)
        main = torch.cuda.current_stream(dev)
        side = main
        # the tail of an overlapped step (phase "b") stays on ONE stream: it is hidden behind the next backbone anyway,
        # and a branch that forks again inside the capture crashed hipStreamEndCapture (ROCm 7.2)
        fork_b = self.side_stream and phase != "b"
        if self.side_stream and phase != "b":
            side = self._side_streams.get(dev.index)
            if side is None:
                side = self._side_streams[dev.index] = torch.cuda.Stream(device=dev)
        pa = {}

        def part_a():
            """Everything that does not read the temporal maps (detection heads, region FC, K / V maps, semantic
            inputs).  With `side_stream` it is forked onto a second stream once residual stage `side_fork_stage` of
            the backbone has been launched, and joined before the temporal branches: the HBM-bound weight-streaming
            FCs then overlap the L2- / latency-bound late backbone stages."""
            if self.side_stream:
                side.wait_stream(main)
            with torch.cuda.stream(side), ops.workspace_slot(1 if self.side_stream else 0):
                # detection heads (:742-743, :938-952).  f32 values that feed a Linear come out of the producing
                # GEMM a second time as bf16 (with the consumer's ReLU / LeakyReLU applied): no cast / act launches
                pooled = ops.roi_align_nhwc(blob, det_all_d, cfg.CONV_BODY_SCALE, ps, cfg.ROI_XFORM_SAMPLING_RATIO)
                pooled = pooled.view(Ntot, ps * ps * C)
                h1 = self._gemm(pooled, "fc1_bso", [self.Box_Head.fc1, self.S_Head.fc1, self.O_Head.fc1], (C, ps * ps), relu=True,
                                tiled=True)
                ori, ori_relu = self._gemm(h1[:, :D], "box_fc2", [self.Box_Head.fc2], out_f32=True, out2="relu")
                w_so, b_so, _ = self._stack_w("so_fc2", [self.S_Head.fc2, self.O_Head.fc2])
                det_so, det_so_bf = ops.gemm_nt(h1.view(Ntot, 3, D)[:, 1:].permute(1, 0, 2), w_so, bias=b_so, out_f32=True,
                                                relu=not cfg.NO_FC7_RELU, out2="none")      # S_Head | O_Head fc2, one launch
                # class scores: the N-padding columns get a -1e30 bias, so the softmax over the padded row is exact
                cls_pad = F.softmax(self._gemm(ori_relu, "cls", [self.Box_Outs.cls_score], out_f32=True, keep_pad=True,
                                               pad_bias=-1e30), dim=1)
                cls = cls_pad[:, :cfg.NUM_CLASSES]
                if "lab" in ix:                                                          # labels / scores came with the boxes (:895-898)
                    det_labels, det_scores = dv[ix["lab"]], dv[ix["sc"]]
                else:
                    det_scores, det_labels = cls[:, 1:].max(dim=1)                       # :903-904
                    det_labels = det_labels + 1
                emb = self._gemm(ops.act_cast(cls_pad), "embeds", [self.embed_obj_det, self.embed_s, self.embed_o], leaky=True)
                det_feat, det_feat_bf = self._gemm(emb[:, :E], "after_emb", [self.after_obj_det_embed[1]], out_f32=True,
                                                   residual=ori, out2="none")             # :945-946
                # regions (:222-245)
                reg_map = ops.roi_align_nhwc(fmap, dv[ix["reg"]], cfg.REL_SPATIAL_SCALE, ps, 0).view(Mtot, ps * ps * C)
                r1 = self._gemm(reg_map, "fc_r0", [ct3.fc_r[0]], (C, ps * ps), relu=True, tiled=True)
                reg_feat, reg_feat_bf = self._gemm(r1, "fc_r2", [ct3.fc_r[2]], out_f32=True, out2="none")
                # K / V maps of the pair attention (:287-289, :1545-1546)
                fm2 = fmap.view(-1, fmap.shape[-1])
                kmap = self._gemm(fm2, "kmap", [self.atten_q_map]).view(B, fmap.shape[1], fmap.shape[2], low)
                if self.fuse_pair_infer:
                    # [V | W_A V | W_B V]: ROIAlign commutes with the 1x1 convs spatio_inferA / B (:1039), so they are
                    # applied to the 14x14 map once per clip instead of to every pair's pooled map
                    def make_vab():
                        wv = self.spatio_atten_map.weight.detach().float().view(low, -1)
                        bv = self.spatio_atten_map.bias.detach().float()
                        wa = self.spatio_inferA[0].weight.detach().float().view(low, low)
                        wb = self.spatio_inferB[0].weight.detach().float().view(low, low)
                        w = torch.cat([wv, wa @ wv, wb @ wv], 0).to(torch.bfloat16).contiguous()
                        b = torch.cat([bv, wa @ bv, wb @ bv], 0).contiguous()
                        bab = (self.spatio_inferA[0].bias.detach().float() + self.spatio_inferB[0].bias.detach().float()).contiguous()
                        return w, b, bab
                    w_vab, b_vab, bias_ab = self._cached("vab", make_vab, self._layer_deps(
                        [self.spatio_atten_map, self.spatio_inferA[0], self.spatio_inferB[0]]))
                    vmap = None
                    if self.pair_mfma and ops.pair_mfma_supported(fmap.shape[1], fmap.shape[2], ps, low):
                        # matrix-core pair kernel: it reads the maps channel-major (the same GEMM run transposed, its bias
                        # applied inside the kernel) and needs no pooled intermediate
                        vab = None
                        pa["vabT"], pa["b_vab"] = ops.pair_maps_channel_major(fmap, w_vab), b_vab
                    else:
                        vab = ops.gemm_nt(fm2, w_vab, bias=b_vab).view(B, fmap.shape[1], fmap.shape[2], 3 * low)
                    # with the second stream the union-box pooling (the heavy half of the pair kernel) runs here,
                    # beside the backbone; what depends on the queries is a streaming pass after the tree-GRU
                    part_d = dv[ix["PART32"]] if self.share_swapped_pairs else None
                    pa["pooled_vab"] = (ops.pair_pool(vab, union_d, cfg.REL_SPATIAL_SCALE, ps, partner=part_d)
                                        if vab is not None and self.side_stream and self.split_pair_pool is True else None)
                else:
                    vmap = self._gemm(fm2, "vmap", [self.spatio_atten_map]).view(B, fmap.shape[1], fmap.shape[2], low)
                # semantic branch inputs that only need the detection heads (:1643-1668, :1066-1069): the subject /
                # object Linears run as 2-batch GEMMs, the per-pair gathers + sums + frequency bias as one kernel
                w_se, b_se, _ = self._stack_w("sem_so_e", [self.sem_s_e[1], self.sem_o_e[1]])
                lows = ops.gemm_nt(emb.view(Ntot, 3, E)[:, 1:].permute(1, 0, 2), w_se, bias=b_se, out_f32=True)
                w_sm, b_sm, n_rel = self._stack_w("sem_so", [self.sem_s[0], self.sem_o[0]])
                parts = ops.gemm_nt(det_so_bf, w_sm, bias=b_sm, out_f32=True)            # (2, Ntot, n_rel padded)
                freq = self.freq_bias.rel_baseline.weight.detach() if cfg.USE_FREQ_BIAS else None    # :1064,:1075
                per_pair = tuple(dv[ix[k]] for k in ("plab_s", "plab_o", "psc_s", "psc_o")) if "plab_s" in ix else None
                sbj_lab_all, obj_lab_all, sbj_sc_all, obj_sc_all, so_bias, so_part = ops.pair_assemble(
                    det_labels, det_scores, X, Y, lows[0], lows[1], parts[0], parts[1], freq, n_rel, cfg.NUM_CLASSES - 1,
                    per_pair=per_pair)
                pa.update(det_labels=det_labels, det_scores=det_scores, det_feat=det_feat, det_feat_bf=det_feat_bf,
                          reg_feat=reg_feat, reg_feat_bf=reg_feat_bf, kmap=kmap, vmap=vmap,
                          vab=vab if self.fuse_pair_infer else None, bias_ab=bias_ab if self.fuse_pair_infer else None,
                          sbj_lab_all=sbj_lab_all, obj_lab_all=obj_lab_all, sbj_sc_all=sbj_sc_all, obj_sc_all=obj_sc_all,
                          so_bias=so_bias, so_part=so_part, n_rel=n_rel)

        # ---- I3D backbone (:978-981)
        split = self.split_stage if phase in ("a", "b") else None
        if phase == "b":
            pa = dict(hand)
            tmaps = hand["tmaps"]
            if split is not None:        # `tmaps` is the activation behind stage `split`: the late stages run here
                tmaps = self.Prd_RCNN.forward_ndhwc(tmaps, start_at=split)
        elif split is not None:
            # part A forks behind `side_fork_stage` if the first half runs that far, else behind its last whole stage
            whole = split if isinstance(split, int) else split[0] - 1
            hook = (min(self.side_fork_stage, whole), part_a) if self.side_stream else None
            if hook is None:
                part_a()
            tmaps = self.Prd_RCNN.forward_ndhwc(all_frames, in_ntchw=True, stage_hook=hook, stop_after=split)
        else:
            hook = None
            if self.side_stream and not graph_backbone:
                hook = (self.side_fork_stage, part_a)
            else:
                part_a()
            if graph_backbone:
                tmaps = self.Prd_RCNN.forward_ndhwc_graphed(all_frames, in_ntchw=True)
            else:
                tmaps = self.Prd_RCNN.forward_ndhwc(all_frames, in_ntchw=True, stage_hook=hook)   # (B,T',h,w,2048) bf16
        det_labels, det_scores, det_feat, det_feat_bf = pa["det_labels"], pa["det_scores"], pa["det_feat"], pa["det_feat_bf"]
        reg_feat, reg_feat_bf, kmap, vmap, vab, bias_ab = (pa["reg_feat"], pa["reg_feat_bf"], pa["kmap"], pa["vmap"],
                                                           pa["vab"], pa["bias_ab"])
        sbj_lab_all, obj_lab_all, sbj_sc_all, obj_sc_all = pa["sbj_lab_all"], pa["obj_lab_all"], pa["sbj_sc_all"], pa["obj_sc_all"]
        so_bias, so_part, n_rel = pa["so_bias"], pa["so_part"], pa["n_rel"]
        if self.side_stream and phase != "b":
            main.wait_stream(side)
        if phase == "a":
            out = dict(pa)
            out["tmaps"] = tmaps
            return out
        Tn = tmaps.shape[1]
        if Tn != prep["Tn"]:
            raise RuntimeError("backbone produced %d frames, host lists were built for %d" % (Tn, prep["Tn"]))
        tm_flat = tmaps.view(B * Tn, tmaps.shape[2], tmaps.shape[3], tmaps.shape[4])

        # ---- cluster_tree3d: temporal object features (:158-201)
        def temporal_branch(rois_d, cur_feat, cur_feat_bf, fc, q_s_l, q_o_l, val, tag):
            tvec = ops.roi_align_avg_nhwc(tm_flat, rois_d, cfg.REL_SPATIAL_SCALE, ps, 0)  # ROIAlign + box_head3d.py:60-61
            tfeat, tfeat_bf = self._gemm(tvec, tag + "_fc2", [fc.fc2], relu=True, out_f32=True, out2="none")  # (K,D), (obj,t) order
            k = cur_feat.shape[0]
            qs = self._gemm(cur_feat_bf, tag + "_qs", [q_s_l[0]], out_f32=True)
            qo = self._gemm(tfeat_bf, tag + "_qo", [q_o_l[0]], out_f32=True)
            att = ops.temporal_attention(qs, qo.view(k, Tn, low), tfeat.view(k, Tn, D), 1.0 / math.sqrt(low), out_bf16=True)
            v = self._gemm(att, tag + "_v0", [val[0]], out_f32=True)
            v = ops.layernorm_act(v, val[1].weight.detach(), val[1].bias.detach(), val[1].eps, "relu")
            # cur_feat + value MLP (:201, :305); second output = the ReLU'd operand of o2c / r2c (:374-379)
            return self._gemm(v, tag + "_v3", [val[3]], out_f32=True, residual=cur_feat, out2="relu")

        if fork_b:
            side.wait_stream(main)
        with torch.cuda.stream(side), ops.workspace_slot(1 if fork_b else 0):
            reg_agg, reg_agg_relu = temporal_branch(dv[ix["t_reg"]], reg_feat, reg_feat_bf, ct3.temporal_r_fc,
                                                    ct3.temporal_r_atten_q_s, ct3.temporal_r_atten_q_o,
                                                    ct3.temporal_r_feat_val, "treg")
            r2c = self._gemm(reg_agg_relu, "r2c", [ct3.r2c[1]], out_f32=True)
            ev_r2c = None
            if (fork_b or os.environ.get("TRACE_TAIL_SPLIT")) and self.side_stream and self.fuse_pair_infer and \
                    self.split_pair_pool == "tail" and vab is not None:
                # the union-box pooling of the pair kernel depends on nothing computed after part A: it runs on the
                # second stream beside the object temporal branch, the tree-GRU (32 workgroups) and the query GEMMs -
                # a chain of small launches that leaves most CUs idle - and only the combine stays on the critical path
                if fork_b:
                    ev_r2c = torch.cuda.Event()
                    ev_r2c.record(side)
                pa["pooled_vab"] = ops.pair_pool(vab, union_d, cfg.REL_SPATIAL_SCALE, ps,
                                                 partner=dv[ix["PART32"]] if self.share_swapped_pairs else None)
        obj_agg, obj_agg_relu = temporal_branch(dv[ix["t_obj"]], det_feat, det_feat_bf, ct3.temporal_obj_fc,
                                                ct3.temporal_obj_atten_q_s, ct3.temporal_obj_atten_q_o,
                                                ct3.temporal_obj_feat_val, "tobj")
        o2c = self._gemm(obj_agg_relu, "o2c", [ct3.o2c[1]], out_f32=True)
        # ---- temporal region features (:268-305): independent of the object branch (a chain of small,
        # latency-bound launches), so with `side_stream` the two chains run side by side
        if fork_b:
            if ev_r2c is not None:
                main.wait_event(ev_r2c)
            else:
                main.wait_stream(side)
        # ---- tree message passing (:307-330)
        fc = ct3.feat_combine
        G = ct3.group_coeffi
        gw = fc // G
        x_nodes = torch.empty((Ntot + Mtot, fc), dtype=torch.float32, device=dev)
        x_nodes[node_obj_d] = o2c
        x_nodes[node_reg_d] = r2c
        h, hd = ops.tree_gru(x_nodes, gw, gw, G, forest, self._tree_weights(), max_tree_nodes=prep["max_tree_nodes"])
        w12, b12 = self._ffn12_w()
        agg = ops.gemm_nt(ops.act_cast(h, None, hd), w12, bias=b12, relu=True)   # FFN1(h)+FFN2(hd) of all groups, ReLU of :387
        pf = self._gemm(agg, "ct_ffn", [ct3.FFN[1]], leaky=True)                 # leaky: only consumers are :273-275, :310-312
        pf_obj, pf_reg = pf[node_obj_d], pf[node_reg_d]
        # ---- back in _forward (:1018-1024)
        obj_feature, obj_feature_relu = self._gemm(pf_obj, "restore", [self.restore_obj_feat[1]], out_f32=True,
                                                   residual=det_feat, out2="relu")
        obj_score = F.softmax(self._gemm(obj_feature_relu, "cem_cls", [self.cem_Box_Outs.cls_score], out_f32=True), dim=1)
        # ---- target-adaptive pair attention (:1524-1612)
        lo = ops.act_cast(obj_feature, "leaky")
        P = union_d.shape[0]
        s_map = o_map = v_map = None
        if self.fuse_pair_infer:
            # [q_s | q_o] in one GEMM (bf16: the operand of the query x K-map product), rows viewed (2*Ntot, low)
            q_so = self._gemm(lo, "att_so", [self.spatio_atten_s[1], self.spatio_atten_o[1]])
            q_s, q_o = q_so[:, :low], q_so[:, low:]
            qk = ops.gemm_nt(q_so.view(2 * Ntot, low), kmap.view(-1, low), out_f32=True)   # (2*Ntot, B*h*w)
            if pa.get("vabT") is not None:
                sp = ops.pair_infer_mfma(pa["vabT"], (fmap.shape[1], fmap.shape[2]), pa["b_vab"], qk, union_d, dv[ix["X32"]],
                                         dv[ix["Y32"]], bias_ab, cfg.REL_SPATIAL_SCALE, ps,
                                         partner=dv[ix["PART32"]] if self.share_swapped_pairs else None, work=dv[ix["PWORK"]])
            elif pa.get("pooled_vab") is not None:
                if ev_r2c is not None:
                    main.wait_stream(side)
                sp = ops.pair_combine(pa["pooled_vab"], tuple(vab.shape[:3]), qk, union_d, dv[ix["X32"]], dv[ix["Y32"]],
                                      bias_ab, cfg.REL_SPATIAL_SCALE, ps,
                                      partner=dv[ix["PART32"]] if self.share_swapped_pairs else None)
            else:
                sp = ops.pair_infer(vab, qk, union_d, dv[ix["X32"]], dv[ix["Y32"]], bias_ab, cfg.REL_SPATIAL_SCALE, ps,
                                    partner=dv[ix["PART32"]] if self.share_swapped_pairs else None)
        else:
            q_s = self._gemm(lo, "att_s", [self.spatio_atten_s[1]], out_f32=True)
            q_o = self._gemm(lo, "att_o", [self.spatio_atten_o[1]], out_f32=True)
            s_map, o_map, v_map, so = ops.pair_attention(kmap, vmap, union_d, dv[ix["X32"]], dv[ix["Y32"]], q_s, q_o,
                                                         cfg.REL_SPATIAL_SCALE, ps)
            so = so.view(P * ps * ps, 2 * low)                                   # [s_map | o_map], written in place
            wab, bab, _ = self._cached("inferAB", lambda: (
                torch.cat([self.spatio_inferA[0].weight.detach().float().view(low, low),
                           self.spatio_inferB[0].weight.detach().float().view(low, low)], 1).to(torch.bfloat16).contiguous(),
                (self.spatio_inferA[0].bias.detach().float() + self.spatio_inferB[0].bias.detach().float()).contiguous(), low),
                self._layer_deps([self.spatio_inferA[0], self.spatio_inferB[0]]))
            sp = ops.gemm_nt(so, wab, bias=bab, leaky=True, residual=v_map.view(P * ps * ps, low), residual_post=True)  # :1039-1040
        spatio_infer = self._gemm(sp.view(P, ps * ps * low), "sp_fc", [self.spatio_infer_fc[0]], (low, ps * ps), leaky=True)  # :1042 (+ leaky of :329)
        # ---- semantic branch (:1643-1668) and the sum of the four logit terms (:1066-1069, + freq bias :1075):
        # each Linear adds the running sum in its epilogue
        low_r = self._gemm(pf_reg, "sem_r", [self.sem_r[1]], out_f32=True)
        sem = ops.gather_add_act(so_bias, low_r, LCA, "leaky")                  # low_s[X] + low_r[lca] + low_o[Y], :326 LeakyReLU
        logit_sem = self._gemm(sem, "rel_sem", [self.relation_infer_sem[2]], out_f32=True, residual=so_part, keep_pad=True)
        logits_pad = self._gemm(spatio_infer, "rel_spa", [self.relation_infer_spa[2]], out_f32=True, residual=logit_sem,
                                keep_pad=True)
        logits = logits_pad[:, :n_rel]
        prd = (torch.sigmoid(logits_pad) if cfg.MULTI_RELATION else F.softmax(logits, dim=1))[:, :n_rel]   # :1083 / :1090

        # ---- gathers and the ranking run once over all clips (:1162-1182 values)
        top = None
        if topk:
            pmax = int(np.max(np.diff(poff))) if len(poff) > 1 else 0
            k_out = min(topk, max(pmax * min(prd_k, prd.shape[1]), 1))
            if not ops.rank_triplets_supported(prd.shape[1], pmax, prd_k, k_out):
                raise RuntimeError("rank_triplets: %d predicate classes / %d pairs x prd_k %d / topk %d exceed the device "
                                   "ranking's limits (see trace_rank_triplets)" % (prd.shape[1], pmax, prd_k, k_out))
            top = ops.rank_triplets(prd, sbj_sc_all, obj_sc_all, sbj_lab_all, obj_lab_all, dv[ix["poff"]], pmax, k_out, prd_k)
        self._debug = dict(det_feat=det_feat, obj_agg=obj_agg, reg_agg=reg_agg, pf=pf, obj_feature=obj_feature,
                           q_s=q_s, q_o=q_o, s_map=s_map, o_map=o_map, v_map=v_map, sp=sp.view(P, ps * ps, low),
                           spatio_infer=spatio_infer,
                           logits=logits, node_obj=node_obj_d, node_reg=node_reg_d, tmaps=tmaps)
        return dict(det_labels=det_labels, det_scores=det_scores, sbj_labels=sbj_lab_all, obj_labels=obj_lab_all,
                    sbj_scores=sbj_sc_all, obj_scores=obj_sc_all, msp_det_scores=obj_score, prd_scores=prd, top=top)

    @staticmethod
    def _make_outs(prep, res, topk, prd_k):
        """Per-clip return dicts (:1162-1182): views into the step's flat tensors + the host-side roi lists."""
        off, poff, det_all, pairs = prep["off"], prep["poff"], prep["det_all"], prep["pairs"]
        top = res["top"]
        nprd = res["prd_scores"].shape[1]
        outs = []
        for b in range(prep["B"]):
            s, o = pairs[b][0], pairs[b][1]
            rois_b = det_all[off[b]:off[b + 1]].copy()
            rois_b[:, 0] = 0
            sl = slice(int(poff[b]), int(poff[b + 1]))
            ol = slice(int(off[b]), int(off[b + 1]))
            d = dict(sbj_rois=rois_b[s], obj_rois=rois_b[o], det_rois=rois_b, sbj_inds=s, obj_inds=o,
                     det_labels=res["det_labels"][ol], det_scores=res["det_scores"][ol],
                     sbj_labels=res["sbj_labels"][sl], obj_labels=res["obj_labels"][sl],
                     sbj_scores=res["sbj_scores"][sl], obj_scores=res["obj_scores"][sl],
                     msp_det_scores=res["msp_det_scores"][ol], prd_scores=res["prd_scores"][sl],
                     prd_ttl_scores=res["prd_scores"][sl])
            if top is not None:
                k = min(topk, (int(poff[b + 1]) - int(poff[b])) * min(prd_k, nprd))
                d.update({name: v[b, :k] for name, v in top.items()})
            outs.append(d)
        return outs
