"""Head modules of the relation model with the reference's names, parameter shapes (so its checkpoints load),
constructor arguments and `forward` signatures, running on libtrace_amd.so:

  roi_2mlp_head, fast_rcnn_outputs   lib/modeling_rel/fast_rcnn_heads.py:94-152, 15-51
  Tail_Layer_3D                      lib/i3d/box_head3d.py:13-79 (I3D_DC5: avg-pool + fc2)
  tree_message_passing2D             lib/combine_tree/cluster_tree.py:70-204
  cluster_tree3d                     lib/combine_tree/cluster_tree3d.py:20-366 (+ cluster_tree2D, cluster_tree.py:355-734)

Initialisers follow the reference (Caffe2 XavierFill / MSRAFill, nn/init.py:11-24; orthogonal, utils_t.py:130-134).

The module-level forwards below are what a caller of the reference's modules gets when it swaps the import: same
inputs (numpy roi blobs, float32 NCHW maps), same outputs (float32 tensors, the 13-tuple of cluster_tree3d.forward).
`Generalized_RCNN` does not call them on its hot path: it runs the same arithmetic fused over all clips of a step
(trace_amd/model_builder_rel.py), reading these modules' parameters.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import cluster_tree as ct
from . import ops
from .config import get_cfg
from .roi_xform import add_rel_multilevel_rois, roi_feature_transform


def xavier_fill(t):  # nn/init.py:11-16
    fan_in = t.numel() / t.shape[0]
    return nn.init.uniform_(t, -math.sqrt(3 / fan_in), math.sqrt(3 / fan_in))


def msra_fill(t):  # nn/init.py:19-24
    fan_out = t.numel() / t.shape[1]
    return nn.init.normal_(t, 0, math.sqrt(2 / fan_out))


def init_xavier(m):  # utils_t.weight_init_mynn_Xavier (utils_t.py:81-88)
    if isinstance(m, (nn.Conv2d, nn.Linear)):
        xavier_fill(m.weight)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)


def init_orthogonal(m):  # utils_t.py:130-134
    if isinstance(m, (nn.Conv2d, nn.Linear)):
        nn.init.orthogonal_(m.weight.data)
        if m.bias is not None:
            m.bias.data.fill_(0.)


def init_msra_conv(m):  # utils_t.py:63-67
    if isinstance(m, (nn.Conv2d, nn.Conv3d)):
        msra_fill(m.weight)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)


def _bf(x):
    return x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16)


class PackedLinears(nn.Module):
    """Mixin: Linear / 1x1-conv layers as operands of the bf16 MFMA GEMM (weights packed once: bf16, K padded to 8,
    N padded to 8, optional column permutation to the channels-last pooled order, several layers stacked along N).
    A packed entry remembers (data_ptr, version) of the tensors it was built from and is rebuilt when they change
    (state-dict load, device move, in-place edit) - whichever module the load went through.  Edits through `.data`
    (`p.data.copy_()`, `p.data.uniform_()`) do not bump the version counter and are NOT seen: write parameters under
    `torch.no_grad()` (`p.copy_()`), as trace_amd.synthetic does, or call `drop_packed()`."""

    def drop_packed(self):
        """Forget every packed weight (and invalidate captured graphs that hold them): after edits through `.data`."""
        self._pk_store().clear()
        self._pk_epoch = getattr(self, "_pk_epoch", 0) + 1

    def _pk_store(self):
        d = self.__dict__.get("_pk")
        if d is None:
            d = self.__dict__["_pk"] = {}
        return d

    @staticmethod
    def _sig(deps):
        return tuple((t.data_ptr(), t._version) for t in deps if t is not None)

    def _cached(self, key, make, deps=None):
        store = self._pk_store()
        ent = store.get(key)
        sig = self._sig(deps) if deps is not None else None
        if ent is not None and ent[1] == sig:
            return ent[0]
        with torch.no_grad():
            v = make()
        if ent is not None:                        # a stale packed tensor may be baked into captured graphs
            self._pk_epoch = getattr(self, "_pk_epoch", 0) + 1
        store[key] = (v, sig)
        return v

    @staticmethod
    def _layer_deps(layers):
        deps = []
        for l in layers:
            deps.append(l.weight)
            deps.append(l.bias)
        return deps

    @staticmethod
    def _pad_k(w, mult=8):
        k = w.shape[1]
        if k % mult:
            w = F.pad(w, (0, mult - k % mult))
        return w

    @staticmethod
    def _pad_n(w, b, mult=8):
        n = w.shape[0]
        if n % mult:
            w = F.pad(w, (0, 0, 0, mult - n % mult))
            if b is not None:
                b = F.pad(b, (0, mult - n % mult))
        return w, b

    def _lin_w(self, key, layers, perm_hw_c=None, pad_bias=0.0):
        """Packed (W bf16 (Np,Kp), bias f32 (Np) | None, N) of one Linear / 1x1 conv or several
        stacked along the output dimension.  `pad_bias` fills the bias of the padding rows (Np > N)."""
        def make():
            ws, bs = [], []
            for l in layers:
                w = l.weight.detach().float().reshape(l.weight.shape[0], -1)
                if perm_hw_c is not None:  # columns (C, ph, pw) -> (ph, pw, C): channels-last pooled rows
                    c, hw = perm_hw_c
                    w = w.view(w.shape[0], c, hw).permute(0, 2, 1).reshape(w.shape[0], -1)
                ws.append(w)
                bs.append(l.bias.detach().float() if l.bias is not None else torch.zeros(w.shape[0], device=w.device))
            w = torch.cat(ws, 0)
            b = torch.cat(bs, 0) if (any(l.bias is not None for l in layers) or pad_bias != 0.0) else None
            n = w.shape[0]
            w, b = self._pad_n(self._pad_k(w), b)
            if b is not None and pad_bias != 0.0:
                b[n:] = pad_bias
            return w.to(torch.bfloat16).contiguous(), (b.contiguous() if b is not None else None), n
        return self._cached(key, make, self._layer_deps(layers))

    def _stack_w(self, key, layers):
        """Same-shaped Linears as ONE batched GEMM operand: (W bf16 (b,Np,Kp), bias f32 (b,Np), N)."""
        def make():
            ws, bs = [], []
            for l in layers:
                w = l.weight.detach().float().reshape(l.weight.shape[0], -1)
                b = l.bias.detach().float() if l.bias is not None else torch.zeros(w.shape[0], device=w.device)
                w, b = self._pad_n(self._pad_k(w), b)
                ws.append(w)
                bs.append(b)
            return torch.stack(ws).to(torch.bfloat16).contiguous(), torch.stack(bs).contiguous(), layers[0].weight.shape[0]
        return self._cached(key, make, self._layer_deps(layers))

    def _gemm(self, x, key, layers, perm_hw_c=None, keep_pad=False, pad_bias=0.0, tiled=False, **kw):
        """x (M,K) -> (M,N) through the bf16 MFMA GEMM; pads K / slices N as needed (keep_pad: keep the
        N-padding columns).  With out2=... returns (y, y2).  tiled: keep the packed weights in the pre-tiled
        weight-streaming layout (ops.tile_weights) - the 100352-wide FCs."""
        store = self._pk_store()
        tkey = key + "/tiled"
        if tiled and tkey in store and store[tkey][1] == self._sig(self._layer_deps(layers)):
            w, b, n = store[tkey][0]
            kw["b_tiled"] = True
        else:
            stale = tkey in store                       # rebuilt after a weight change: the old tiled copy may be
            w, b, n = self._lin_w(key, layers, perm_hw_c, pad_bias)   # baked into captured step graphs
            if stale:
                self._pk_epoch = getattr(self, "_pk_epoch", 0) + 1
            if tiled and w.shape[0] % 128 == 0 and w.shape[1] % 64 == 0:
                w = ops.tile_weights(w)
                store[tkey] = ((w, b, n), self._sig(self._layer_deps(layers)))
                del store[key]                      # only the tiled copy stays resident
                kw["b_tiled"] = True
        x = _bf(x)
        if x.shape[-1] != w.shape[1]:
            x = F.pad(x, (0, w.shape[1] - x.shape[-1]))
        if x.stride(-1) != 1:
            x = x.contiguous()
        y = ops.gemm_nt(x, w, bias=b, **kw)
        if keep_pad or w.shape[0] == n:
            return y
        if isinstance(y, tuple):
            return tuple(t[..., :n] for t in y)
        return y[..., :n]


def _pooled_rows(y):
    """Pooled roi maps -> (rows (K, ph*pw*C) bf16 channels-last, (C, ph*pw)); accepts (K,C,ph,pw) f32 (the reference
    layout; the `batch, T` leading dims of Tail_Layer_3D inputs are flattened) or (K,ph,pw,C) bf16."""
    if y.dtype == torch.bfloat16:
        K, ph, pw, C = y.shape
        return y.reshape(K, ph * pw * C), (C, ph * pw)
    y = y.reshape(-1, y.shape[-3], y.shape[-2], y.shape[-1])
    K, C, ph, pw = y.shape
    return y.permute(0, 2, 3, 1).reshape(K, ph * pw * C).to(torch.bfloat16), (C, ph * pw)


class roi_2mlp_head(PackedLinears):
    """fast_rcnn_heads.py:94-152.  `roi_2mlp_head(dim_in, roi_xform_func, spatial_scale)` as the reference, or
    `roi_2mlp_head(dim_in, cfg)` / `roi_2mlp_head(dim_in, cfg=cfg)` with an explicit TraceConfig."""

    def __init__(self, dim_in, roi_xform_func=None, spatial_scale=None, cfg=None):
        super().__init__()
        if roi_xform_func is not None and not callable(roi_xform_func):   # (dim_in, cfg) form
            cfg, roi_xform_func = roi_xform_func, None
        cfg = cfg or get_cfg()
        self.cfg = cfg
        self.dim_in = dim_in
        self.roi_xform = roi_xform_func or roi_feature_transform
        self.spatial_scale = cfg.CONV_BODY_SCALE if spatial_scale is None else spatial_scale
        self.dim_out = cfg.MLP_HEAD_DIM
        self.fc1 = nn.Linear(dim_in * cfg.ROI_XFORM_RESOLUTION ** 2, self.dim_out)
        self.fc2 = nn.Linear(self.dim_out, self.dim_out)
        for fc in (self.fc1, self.fc2):
            xavier_fill(fc.weight)
            nn.init.constant_(fc.bias, 0)

    def detectron_weight_mapping(self):   # fast_rcnn_heads.py:115-122
        return {'fc1.weight': 'fc6_w', 'fc1.bias': 'fc6_b', 'fc2.weight': 'fc7_w', 'fc2.bias': 'fc7_b'}, []

    def forward(self, x, rpn_ret, rois_name='rois', use_relu=True):
        return self.map2vec(self.roi2map(x, rpn_ret, rois_name=rois_name), use_relu=use_relu)

    def roi2map(self, x, rpn_ret, rois_name='rois'):
        cfg = self.cfg
        return self.roi_xform(x, rpn_ret, blob_rois=rois_name, method='RoIAlign', resolution=cfg.ROI_XFORM_RESOLUTION,
                              spatial_scale=self.spatial_scale, sampling_ratio=cfg.ROI_XFORM_SAMPLING_RATIO)

    def map2vec(self, x, use_relu=True):
        """(K,C,7,7) f32 or (K,7,7,C) bf16 pooled maps -> (K, dim_out) f32: fc1 + ReLU, fc2 (+ ReLU)."""
        with torch.no_grad():
            rows, perm = _pooled_rows(x)
            if rows.shape[0] == 0:
                return torch.zeros((0, self.dim_out), dtype=torch.float32, device=x.device)
            h = self._gemm(rows, "fc1", [self.fc1], perm, relu=True)
            return self._gemm(h, "fc2", [self.fc2], relu=bool(use_relu), out_f32=True)


class fast_rcnn_outputs(PackedLinears):
    """fast_rcnn_heads.py:15-51.  `fast_rcnn_outputs(dim_in)` (global cfg) or `fast_rcnn_outputs(dim_in, cfg)`."""

    def __init__(self, dim_in, cfg=None, with_bbox=True):
        super().__init__()
        cfg = cfg or get_cfg()
        self.cls_score = nn.Linear(dim_in, cfg.NUM_CLASSES)
        nn.init.normal_(self.cls_score.weight, std=0.01)
        nn.init.constant_(self.cls_score.bias, 0)
        if with_bbox:
            self.bbox_pred = nn.Linear(dim_in, 4 * cfg.NUM_CLASSES)
            nn.init.normal_(self.bbox_pred.weight, std=0.001)
            nn.init.constant_(self.bbox_pred.bias, 0)

    def detectron_weight_mapping(self):   # fast_rcnn_heads.py:32-40
        return {'cls_score.weight': 'cls_score_w', 'cls_score.bias': 'cls_score_b', 'bbox_pred.weight': 'bbox_pred_w',
                'bbox_pred.bias': 'bbox_pred_b'}, []

    def forward(self, x, use_softmax=True):
        """x (K, dim_in) f32 -> (cls_score (K,C) [softmax in eval mode], bbox_pred (K,4C) | None), f32."""
        if x.dim() == 4:
            x = x.squeeze(3).squeeze(2)
        with torch.no_grad():
            cls = self._gemm(x, "cls", [self.cls_score], out_f32=True)
            if not self.training and use_softmax:
                cls = F.softmax(cls, dim=1)
            bbox = self._gemm(x, "bbox", [self.bbox_pred], out_f32=True) if hasattr(self, "bbox_pred") else None
        return cls, bbox


class Tail_Layer_3D(PackedLinears):
    """box_head3d.py:13-79 with I3D_DC5 (no tail res-layers: num_stages == 4; fc1 = AdaptiveAvgPool2d).
    `Tail_Layer_3D(net, net_layer_names, roi_xform_func, spatial_scale, num_stages)` as the reference (`net` needs
    `.final_feat_dim`), or `Tail_Layer_3D(dim_in, cfg)`."""

    def __init__(self, net, net_layer_names=(), roi_xform_func=None, spatial_scale=None, num_stages=4, cfg=None):
        super().__init__()
        if net_layer_names is not None and not isinstance(net_layer_names, (list, tuple)):   # (dim_in, cfg) form
            cfg, net_layer_names = net_layer_names, ()
        cfg = cfg or get_cfg()
        if not cfg.I3D_DC5 or len(list(net_layer_names)[num_stages:]) > 0:
            raise NotImplementedError("Tail_Layer_3D: only the I3D_DC5 form (no tail res-layers) is built")
        self.dim_in = net if isinstance(net, int) else net.final_feat_dim
        self.layer_names = []
        self.spatial_scale = spatial_scale
        self.dim_out = cfg.MLP_HEAD_DIM
        self.fc1 = nn.AdaptiveAvgPool2d((1, 1))
        self.fc2 = nn.Linear(self.dim_in, self.dim_out)
        xavier_fill(self.fc2.weight)
        nn.init.constant_(self.fc2.bias, 0)

    def map2vec(self, x, use_relu=True):
        """(n,T,C,P,P) or (k,C,P,P) f32 pooled maps -> (n*T, dim_out) f32: mean over the bins, fc2 (+ ReLU)."""
        with torch.no_grad():
            x = x.reshape(-1, x.shape[-3], x.shape[-2], x.shape[-1])
            v = x.float().mean(dim=(2, 3))
            return self._gemm(v, "fc2", [self.fc2], relu=bool(use_relu), out_f32=True)

    def forward(self, x, use_relu=True):
        return self.map2vec(x, use_relu=use_relu)


def _seq_linear(i, o, bias):
    return nn.Sequential(nn.Linear(i, o, bias=bias))


class tree_message_passing2D(nn.Module):
    """Parameter holder of cluster_tree.py:70-161; the message passing itself is trace_tree_gru over all groups at
    once (cluster_tree3d.forward / Generalized_RCNN)."""

    def __init__(self, feat_vec_dim, hidden_dim, mode='tree-gru'):
        super().__init__()
        if mode != 'tree-gru':
            raise NotImplementedError("only msp_mode='tree-gru' (TRACE's configuration) is built")
        self.hidden_dim = hidden_dim
        self.feat_vec_dim = feat_vec_dim
        for sfx in ("", "d"):
            setattr(self, "Wh" + sfx, _seq_linear(feat_vec_dim, hidden_dim, False))
            setattr(self, "Wz" + sfx, _seq_linear(feat_vec_dim, hidden_dim, False))
            setattr(self, "Wr" + sfx, _seq_linear(feat_vec_dim, hidden_dim, False))
            setattr(self, "Uh" + sfx, _seq_linear(hidden_dim, hidden_dim, False))
            setattr(self, "Uz" + sfx, _seq_linear(hidden_dim, hidden_dim, True))
            setattr(self, "Ur" + sfx, _seq_linear(hidden_dim, hidden_dim, True))
        self.FFN1 = _seq_linear(hidden_dim, feat_vec_dim, True)
        self.FFN2 = _seq_linear(hidden_dim, feat_vec_dim, True)
        for n, m in self.named_children():
            m.apply(init_xavier if n.startswith("FFN") else init_orthogonal)


def tree_gru_weights(tms):
    """The 16 transposed f32 matrices / biases per group that trace_tree_gru takes."""
    out = []
    for tm in tms:
        d = {}
        for sfx in ("", "d"):
            for n in ("Wz", "Wr", "Wh", "Uz", "Ur", "Uh"):
                d[n + sfx] = getattr(tm, n + sfx)[0].weight.detach().float().t().contiguous()
            d["bz" + sfx] = getattr(tm, "Uz" + sfx)[0].bias.detach().float().contiguous()
            d["br" + sfx] = getattr(tm, "Ur" + sfx)[0].bias.detach().float().contiguous()
        out.append(d)
    return out


def ffn12_weights(tms):
    """FFN1(h) + FFN2(hd) of every group (cluster_tree.py:203) as ONE GEMM over [h | hd]: block-diagonal
    weight (fc, 2*fc) - group g reads h[:, g] and hd[:, g] only - with bias b1 + b2."""
    G = len(tms)
    gw, hd = tms[0].FFN1[0].weight.shape
    w = torch.zeros((G * gw, 2 * G * hd), device=tms[0].FFN1[0].weight.device)
    bs = []
    for g, tm in enumerate(tms):
        w[g * gw:(g + 1) * gw, g * hd:(g + 1) * hd] = tm.FFN1[0].weight.detach().float()
        w[g * gw:(g + 1) * gw, G * hd + g * hd:G * hd + (g + 1) * hd] = tm.FFN2[0].weight.detach().float()
        bs.append(tm.FFN1[0].bias.detach().float() + tm.FFN2[0].bias.detach().float())
    return w.to(torch.bfloat16).contiguous(), torch.cat(bs).contiguous()


class cluster_tree3d(PackedLinears):
    """cluster_tree3d.py:20-366 on top of cluster_tree2D (cluster_tree.py:355-734).

    `cluster_tree3d(feat_map_shape, feat_combine, feat_vec_dim, feat_region_vec_dim, hidden_dim, low_feat_vec_dim,
    Roialign, spatial_scale, group_coeffi, max_tree_height, pool_size, BoxHead, cluster_num_partition, msp_mode)` as
    the reference (`BoxHead` = the Tail_Layer_3D that is deep-copied into temporal_obj_fc / temporal_r_fc), or
    `cluster_tree3d(C, cfg)` with a TraceConfig."""

    def __init__(self, feat_map_shape, feat_combine=None, feat_vec_dim=None, feat_region_vec_dim=None, hidden_dim=None,
                 low_feat_vec_dim=None, Roialign=None, spatial_scale=None, group_coeffi=1, max_tree_height=7,
                 pool_size=5, BoxHead=None, cluster_num_partition=2, msp_mode='tree-gru', cfg=None):
        super().__init__()
        import copy
        if feat_combine is not None and not isinstance(feat_combine, int):        # (C, cfg) form
            cfg, feat_combine = feat_combine, None
        if feat_combine is None:
            cfg = cfg or get_cfg()
            C = feat_map_shape if isinstance(feat_map_shape, int) else feat_map_shape[0]
            D = cfg.MLP_HEAD_DIM
            feat_combine = hidden_dim = D // cfg.RESIZE_HIDDEN_DIM         # model_builder_rel.py:202-205
            feat_vec_dim = feat_region_vec_dim = D
            low_feat_vec_dim = cfg.LOW_FEAT_DIM
            spatial_scale = cfg.REL_SPATIAL_SCALE
            group_coeffi, max_tree_height = cfg.MSP_GROUP, cfg.MSP_MAX_TREE_HEIGHT
            pool_size, cluster_num_partition = cfg.ROI_XFORM_RESOLUTION, cfg.CLUSTER_NUM_PARTITION
            BoxHead = Tail_Layer_3D(C, cfg)
        else:
            C = feat_map_shape[0]
            if BoxHead is None:
                raise NotImplementedError("cluster_tree3d: only the BoxHead (Tail_Layer_3D) form TRACE builds is supported")
            feat_region_vec_dim = BoxHead.dim_out                            # cluster_tree3d.py:22-23
        if msp_mode != 'tree-gru':
            raise NotImplementedError("only msp_mode='tree-gru' (TRACE's configuration) is built")
        D, Dr, fc, low, ps, G = feat_vec_dim, feat_region_vec_dim, feat_combine, low_feat_vec_dim, pool_size, group_coeffi
        self.cluster_num_partition, self.max_tree_height = cluster_num_partition, max_tree_height
        self.pool_size, self.spatial_scale = ps, spatial_scale
        self.feat_vec_dim, self.feat_combine, self.group_coeffi, self.msp_mode = D, fc, G, msp_mode
        self.low_dim = low
        self.sqrt_low_dim_embed_dim = math.sqrt(low)
        self.ROIalign_region = self.temporal_ROIalign_obj = self.temporal_ROIalign_region = self.roi_feature_transform
        # cluster_tree2D (:369-389)
        self.fc_r = nn.Sequential(nn.Linear(C * ps * ps, Dr), nn.ReLU(inplace=True), nn.Linear(Dr, Dr))
        self.r2c = nn.Sequential(nn.ReLU(inplace=True), nn.Linear(Dr, fc, bias=False))
        self.o2c = nn.Sequential(nn.ReLU(inplace=True), nn.Linear(D, fc, bias=False))
        self.tree_message_passing_list = nn.ModuleList(
            [tree_message_passing2D(fc // G, hidden_dim // G, mode=msp_mode) for _ in range(G)])
        self.FFN = nn.Sequential(nn.ReLU(inplace=True), nn.Linear(fc, fc, bias=True))
        # cluster_tree3d (:48-95): BoxHead given -> deep copies of the 3-D tail
        self.use_pretrain_BoxHead = True
        self.temporal_obj_fc = copy.deepcopy(BoxHead)
        self.temporal_r_fc = copy.deepcopy(BoxHead)
        Dt = BoxHead.dim_out
        self.temporal_obj_atten_q_s = _seq_linear(D, low, False)
        self.temporal_obj_atten_q_o = _seq_linear(Dt, low, False)
        self.temporal_obj_feat_val = nn.Sequential(nn.Linear(Dt, D // 4, bias=False), nn.LayerNorm(D // 4),
                                                   nn.ReLU(inplace=True), nn.Linear(D // 4, D, bias=False))
        self.temporal_r_atten_q_s = _seq_linear(Dr, low, False)
        self.temporal_r_atten_q_o = _seq_linear(Dr, low, False)
        self.temporal_r_feat_val = nn.Sequential(nn.Linear(Dr, Dr // 4, bias=False), nn.LayerNorm(Dr // 4),
                                                 nn.ReLU(inplace=True), nn.Linear(Dr // 4, Dr, bias=False))
        for m in (self.fc_r, self.FFN, self.r2c, self.o2c, self.temporal_obj_atten_q_s, self.temporal_obj_atten_q_o,
                  self.temporal_obj_feat_val, self.temporal_r_atten_q_s, self.temporal_r_atten_q_o,
                  self.temporal_r_feat_val):
            m.apply(init_xavier)

    # ------------------------------------------------------------------ reference helpers kept for callers
    @staticmethod
    def roi_feature_transform(blobs_in, rpn_ret, blob_rois='rois', method='RoIAlign', resolution=7,
                              spatial_scale=1. / 16., sampling_ratio=0):
        """cluster_tree.py:689-734 (its own copy of the model's roi_feature_transform; sampling_ratio 0)."""
        return roi_feature_transform(blobs_in, rpn_ret, blob_rois, method, resolution, spatial_scale, sampling_ratio)

    @staticmethod
    def _add_rel_multilevel_rois(blobs):
        return add_rel_multilevel_rois(blobs, 'rois')

    @staticmethod
    def get_bbox_id(roi, batch_size):
        """cluster_tree.py:651-687: per-clip box counts from the batch-index column (boxes sorted by clip)."""
        ids = np.asarray(roi[:, 0]).astype(np.int64)
        counts = [int((ids == b).sum()) for b in range(batch_size)]
        return ids, counts, [0] + np.cumsum(counts).tolist()

    def temporal_obj_attention(self, s, o):
        """cluster_tree3d.py:337-351: s (n,K), o (n,T,K) f32 on the GPU -> (n,K) f32."""
        return self._temporal_attention(s, o, self.temporal_obj_atten_q_s, self.temporal_obj_atten_q_o, "tobj")

    def temporal_r_attention(self, s, o):
        """cluster_tree3d.py:353-366."""
        return self._temporal_attention(s, o, self.temporal_r_atten_q_s, self.temporal_r_atten_q_o, "treg")

    def _temporal_attention(self, s, o, q_s_l, q_o_l, tag, out_bf16=False):
        with torch.no_grad():
            n, T, K = o.shape
            qs = self._gemm(s, tag + "_qs", [q_s_l[0]], out_f32=True)
            qo = self._gemm(o.reshape(n * T, K), tag + "_qo", [q_o_l[0]], out_f32=True)
            return ops.temporal_attention(qs, qo.view(n, T, -1), o.float().contiguous(), 1.0 / self.sqrt_low_dim_embed_dim,
                                          out_bf16=out_bf16)

    # ------------------------------------------------------------------ forward (cluster_tree3d.py:114-335)
    @staticmethod
    def _nhwc(x):
        """(B,C,H,W) f32 (the reference layout) or (B,H,W,C) bf16 -> (B,H,W,C) bf16."""
        if x.dtype == torch.bfloat16:
            return x.contiguous()
        return x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16)

    def _temporal_branch(self, tm_flat, boxes_per_clip, counts, T, cur_feat, tail, q_s_l, q_o_l, val, tag):
        """Temporal ROIAlign + Tail_Layer_3D + attention + value MLP + residual for every clip.
        boxes_per_clip[b] (k_b, T, 4) per-frame boxes.  The reference pairs an OBJECT-major box list with FRAME-major
        frame ids (:158-162) and views the pooled rows (T, k) (:181-182); reproduced as is."""
        dev = tm_flat.device
        rois, gather = [], []
        base = 0
        for b, bx in enumerate(boxes_per_clip):
            k = counts[b]
            flat = np.asarray(bx, np.float32).reshape(-1, 4)                             # object-major (k*T, 4)
            ids = np.repeat(np.arange(b * T, (b + 1) * T), k).astype(np.float32)[:, None]  # frame-major ids
            rois.append(np.concatenate([ids, flat], 1))
            j, t = np.meshgrid(np.arange(k), np.arange(T), indexing="ij")
            gather.append((base + t * k + j).reshape(-1))                                # list row of entry (j, t)
            base += k * T
        rois = torch.from_numpy(np.concatenate(rois, 0).astype(np.float32)).to(dev)
        gather = torch.from_numpy(np.concatenate(gather)).to(dev)
        tvec = ops.roi_align_avg_nhwc(tm_flat, rois, float(self.spatial_scale), self.pool_size, 0)[gather]   # (sum k*T, C)
        tfeat = self._gemm(tvec, tag + "_fc2", [tail.fc2], relu=True, out_f32=True)
        ktot = cur_feat.shape[0]
        att = self._temporal_attention(cur_feat, tfeat.view(ktot, T, -1), q_s_l, q_o_l, tag, out_bf16=True)
        v = self._gemm(att, tag + "_v0", [val[0]], out_f32=True)
        v = ops.layernorm_act(v, val[1].weight.detach(), val[1].bias.detach(), val[1].eps, "relu")
        return self._gemm(v, tag + "_v3", [val[3]], out_f32=True, residual=cur_feat.float().contiguous())

    def forward(self, roi, feat_map, obj_feat_vec, temporal_roi, temporal_feat_map, temporal_obj_feat_map,
                sbj_id=None, obj_id=None):
        """roi (n,5) numpy [clip, x1,y1,x2,y2] sorted by clip; feat_map (B,C,H,W) f32 (or (B,H,W,C) bf16);
        obj_feat_vec (n,K) f32 GPU; temporal_roi (n,T,5) numpy; temporal_feat_map (B,T,C,H,W) f32 (or (B,T,H,W,C) bf16).
        Returns the reference's 13-tuple (:332-335)."""
        if isinstance(feat_map, (list, tuple)) or isinstance(temporal_feat_map, (list, tuple)):
            raise NotImplementedError("cluster_tree3d.forward: FPN feature lists are not used by TRACE's configurations")
        if self.training:
            raise NotImplementedError("trace_amd implements the inference path; call .eval()")
        with torch.no_grad():
            roi = np.asarray(roi, np.float32)
            temporal_roi = np.asarray(temporal_roi, np.float32)
            B = feat_map.shape[0]
            T = temporal_feat_map.shape[1]
            dev = obj_feat_vec.device
            _, n_list, n_sum = self.get_bbox_id(roi, B)
            if temporal_obj_feat_map.dtype == torch.bfloat16:
                tm_flat = temporal_obj_feat_map.reshape((B * T,) + tuple(temporal_obj_feat_map.shape[2:])).contiguous()
            else:
                tm_flat = self._nhwc(temporal_obj_feat_map.reshape((B * T,) + tuple(temporal_obj_feat_map.shape[2:])))
            obj_feat_vec = obj_feat_vec.float().contiguous()
            # ---- temporal object features (:158-201)
            t_obj = [temporal_roi[n_sum[b]:n_sum[b + 1], :, 1:5] for b in range(B)]
            obj_agg = self._temporal_branch(tm_flat, t_obj, n_list, T, obj_feat_vec, self.temporal_obj_fc,
                                            self.temporal_obj_atten_q_s, self.temporal_obj_atten_q_o,
                                            self.temporal_obj_feat_val, "tobj")
            # ---- box trees (host numpy as in the reference, :204-231)
            trees = [ct.build_tree(roi[n_sum[b]:n_sum[b + 1], 1:5], self.max_tree_height, self.cluster_num_partition,
                                   with_layers=True) for b in range(B)]
            layer_lists = [t["layers"] for t in trees]
            tree_n = [int(t["parent"].shape[0]) for t in trees]
            tree_sum = [0] + np.cumsum(tree_n).tolist()
            m_list = [tree_n[b] - n_list[b] for b in range(B)]
            m_sum = [0] + np.cumsum(m_list).tolist()
            A_l, AT_l, link_l = [], [], []
            for b in range(B):
                A, A_T, link = ct.adjacency_and_link(tree_n[b], layer_lists[b])
                A_l.append(A)
                AT_l.append(A_T)
                link_l.append(link)
            non_leaf = np.concatenate([np.concatenate([np.full((m_list[b], 1), b, np.float32),
                                                       trees[b]["boxes"][n_list[b]:].astype(np.float32)], 1)
                                       for b in range(B)], 0).astype(np.float32)
            # ---- region features (:234-245)
            region_map = self.ROIalign_region(feat_map, {'rois': non_leaf}, resolution=self.pool_size,
                                              spatial_scale=self.spatial_scale)
            rows, perm = _pooled_rows(region_map)
            r1 = self._gemm(rows, "fc_r0", [self.fc_r[0]], perm, relu=True)
            reg_feat = self._gemm(r1, "fc_r2", [self.fc_r[2]], out_f32=True)
            # ---- temporal region features (:247-305): per-frame hull of each region's sons
            t_reg = [ct.temporal_hulls(trees[b], t_obj[b]) for b in range(B)]
            reg_agg = self._temporal_branch(tm_flat, t_reg, m_list, T, reg_feat, self.temporal_r_fc,
                                            self.temporal_r_atten_q_s, self.temporal_r_atten_q_o,
                                            self.temporal_r_feat_val, "treg")
            # ---- tree message passing (:307-330)
            r2c = self._gemm(ops.act_cast(reg_agg, "relu"), "r2c", [self.r2c[1]], out_f32=True)
            o2c = self._gemm(ops.act_cast(obj_agg, "relu"), "o2c", [self.o2c[1]], out_f32=True)
            fc, G = self.feat_combine, self.group_coeffi
            x_nodes = torch.empty((tree_sum[-1], fc), dtype=torch.float32, device=dev)
            for b in range(B):
                x_nodes[tree_sum[b]:tree_sum[b] + n_list[b]] = o2c[n_sum[b]:n_sum[b + 1]]
                x_nodes[tree_sum[b] + n_list[b]:tree_sum[b + 1]] = r2c[m_sum[b]:m_sum[b + 1]]
            forest = {k: torch.from_numpy(v).to(dev) for k, v in ct.forest_arrays(trees).items()}
            tms = self.tree_message_passing_list
            tw = self._cached("tree", lambda: tree_gru_weights(tms), [p for tm in tms for p in tm.parameters()])
            h, hd = ops.tree_gru(x_nodes, fc // G, tms[0].hidden_dim, G, forest, tw, max_tree_nodes=max(tree_n))
            w12, b12 = self._cached("ffn12", lambda: ffn12_weights(tms), [p for tm in tms for p in tm.parameters()])
            agg = ops.gemm_nt(ops.act_cast(h, None, hd), w12, bias=b12, relu=True)       # FFN1(h)+FFN2(hd), ReLU of :387
            pooled_features = self._gemm(agg, "ct_ffn", [self.FFN[1]], out_f32=True)      # :330
            if region_map.dtype == torch.bfloat16:        # reference layout for the caller
                region_map = region_map.permute(0, 3, 1, 2).float()
        return (pooled_features, region_map, layer_lists, tree_n, tree_sum, n_list, n_sum, m_list, m_sum,
                A_l, AT_l, link_l, non_leaf)
