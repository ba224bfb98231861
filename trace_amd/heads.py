"""Parameter containers of the relation head, named and shaped as in the reference so its
checkpoints load:

  roi_2mlp_head, fast_rcnn_outputs   lib/modeling_rel/fast_rcnn_heads.py:94-152, 15-51
  Tail_Layer_3D                      lib/i3d/box_head3d.py:13-79 (I3D_DC5: avg-pool + fc2)
  tree_message_passing2D             lib/combine_tree/cluster_tree.py:70-204
  cluster_tree3d                     lib/combine_tree/cluster_tree3d.py:20-111 (+ cluster_tree2D :355-398)

Initialisers follow the reference (Caffe2 XavierFill / MSRAFill, nn/init.py:11-24; orthogonal,
utils_t.py:130-134).  The compute lives in trace_amd/model_builder_rel.py.
"""
import math

import torch
import torch.nn as nn


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


class roi_2mlp_head(nn.Module):
    def __init__(self, dim_in, cfg):
        super().__init__()
        self.dim_in = dim_in
        self.dim_out = cfg.MLP_HEAD_DIM
        self.fc1 = nn.Linear(dim_in * cfg.ROI_XFORM_RESOLUTION ** 2, self.dim_out)
        self.fc2 = nn.Linear(self.dim_out, self.dim_out)
        for fc in (self.fc1, self.fc2):
            xavier_fill(fc.weight)
            nn.init.constant_(fc.bias, 0)


class fast_rcnn_outputs(nn.Module):
    def __init__(self, dim_in, cfg, with_bbox=True):
        super().__init__()
        self.cls_score = nn.Linear(dim_in, cfg.NUM_CLASSES)
        nn.init.normal_(self.cls_score.weight, std=0.01)
        nn.init.constant_(self.cls_score.bias, 0)
        if with_bbox:
            self.bbox_pred = nn.Linear(dim_in, 4 * cfg.NUM_CLASSES)
            nn.init.normal_(self.bbox_pred.weight, std=0.001)
            nn.init.constant_(self.bbox_pred.bias, 0)


class Tail_Layer_3D(nn.Module):
    """box_head3d.py:13-46 with I3D_DC5 (no tail res-layers, fc1 = AdaptiveAvgPool2d)."""

    def __init__(self, dim_in, cfg):
        super().__init__()
        self.dim_in = dim_in
        self.dim_out = cfg.MLP_HEAD_DIM
        self.fc1 = nn.AdaptiveAvgPool2d((1, 1))
        self.fc2 = nn.Linear(dim_in, self.dim_out)
        xavier_fill(self.fc2.weight)
        nn.init.constant_(self.fc2.bias, 0)


def _seq_linear(i, o, bias):
    return nn.Sequential(nn.Linear(i, o, bias=bias))


class tree_message_passing2D(nn.Module):
    def __init__(self, feat_vec_dim, hidden_dim):
        super().__init__()
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


class cluster_tree3d(nn.Module):
    def __init__(self, C, cfg):
        super().__init__()
        D = cfg.MLP_HEAD_DIM
        fc = D // cfg.RESIZE_HIDDEN_DIM          # feat_combine == hidden_dim (model_builder_rel.py:202-205)
        low = cfg.LOW_FEAT_DIM
        ps = cfg.ROI_XFORM_RESOLUTION
        G = cfg.MSP_GROUP
        self.feat_combine, self.group_coeffi, self.pool_size = fc, G, ps
        # cluster_tree2D (:369-389)
        self.fc_r = nn.Sequential(nn.Linear(C * ps * ps, D), nn.ReLU(inplace=True), nn.Linear(D, D))
        self.r2c = nn.Sequential(nn.ReLU(inplace=True), nn.Linear(D, fc, bias=False))
        self.o2c = nn.Sequential(nn.ReLU(inplace=True), nn.Linear(D, fc, bias=False))
        self.tree_message_passing_list = nn.ModuleList(
            [tree_message_passing2D(fc // G, fc // G) for _ in range(G)])
        self.FFN = nn.Sequential(nn.ReLU(inplace=True), nn.Linear(fc, fc, bias=True))
        # cluster_tree3d (:48-95): BoxHead given -> deep copies of the 3-D tail
        self.temporal_obj_fc = Tail_Layer_3D(C, cfg)
        self.temporal_r_fc = Tail_Layer_3D(C, cfg)
        self.temporal_obj_atten_q_s = _seq_linear(D, low, False)
        self.temporal_obj_atten_q_o = _seq_linear(D, low, False)
        self.temporal_obj_feat_val = nn.Sequential(nn.Linear(D, D // 4, bias=False), nn.LayerNorm(D // 4),
                                                   nn.ReLU(inplace=True), nn.Linear(D // 4, D, bias=False))
        self.temporal_r_atten_q_s = _seq_linear(D, low, False)
        self.temporal_r_atten_q_o = _seq_linear(D, low, False)
        self.temporal_r_feat_val = nn.Sequential(nn.Linear(D, D // 4, bias=False), nn.LayerNorm(D // 4),
                                                 nn.ReLU(inplace=True), nn.Linear(D // 4, D, bias=False))
        for m in (self.fc_r, self.FFN, self.r2c, self.o2c, self.temporal_obj_atten_q_s, self.temporal_obj_atten_q_o,
                  self.temporal_obj_feat_val, self.temporal_r_atten_q_s, self.temporal_r_atten_q_o,
                  self.temporal_r_feat_val):
            m.apply(init_xavier)
