"""RPN heads and proposal generation on the device.

Reference: Detectron_pytorch/lib/modeling/rpn_heads.py `single_scale_rpn_outputs` (:39-130), generate_anchors.py
(:48-125), generate_proposals.py `GenerateProposalsOp` (:12-200).  The reference copies the score / delta maps to the
host and runs numpy + Cython NMS there; here the 3x3 / 1x1 convs go through the implicit-GEMM family, and the proposal
op stays on the GPU (stable sort, HIP box decode, device NMS) until the final <= post_nms_topN rois, which are
returned as ndarrays like the reference's.  Module / parameter names are the reference's (RPN_conv, RPN_cls_score,
RPN_bbox_pred), so its checkpoints load.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import boxes as box_ops
from . import _C, ops
from .resnet3d import _PACK_EPOCH


def generate_anchors(stride=16, sizes=(32, 64, 128, 256, 512), aspect_ratios=(0.5, 1, 2)):
    """generate_anchors.py:48-125: (A,4) float64, aspect ratios outer, scales inner."""
    base = float(stride)
    scales = np.asarray(sizes, np.float64) / stride
    ratios = np.asarray(aspect_ratios, np.float64)

    def whctrs(a):
        w, h = a[2] - a[0] + 1, a[3] - a[1] + 1
        return w, h, a[0] + 0.5 * (w - 1), a[1] + 0.5 * (h - 1)

    def mkanchors(ws, hs, xc, yc):
        ws, hs = ws[:, np.newaxis], hs[:, np.newaxis]
        return np.hstack((xc - 0.5 * (ws - 1), yc - 0.5 * (hs - 1), xc + 0.5 * (ws - 1), yc + 0.5 * (hs - 1)))

    w, h, xc, yc = whctrs(np.array([1, 1, base, base], np.float64) - 1)
    ws = np.round(np.sqrt(w * h / ratios))
    rows = mkanchors(ws, np.round(ws * ratios), xc, yc)
    out = []
    for r in rows:
        w, h, xc, yc = whctrs(r)
        out.append(mkanchors(w * scales, h * scales, xc, yc))
    return np.vstack(out)


class GenerateProposalsOp(nn.Module):
    """generate_proposals.py:12-200.  TEST-mode settings (the path is inference): pre / post NMS top-N, NMS threshold,
    minimum size, as configs/vidvrd/*dc5_2d_new.yaml:68-69 and lib/core/config.py:227-240."""

    def __init__(self, anchors, spatial_scale, pre_nms_topN=6000, post_nms_topN=1000, nms_thresh=0.7, min_size=0):
        super().__init__()
        self._anchors = np.asarray(anchors, np.float64)
        self._num_anchors = self._anchors.shape[0]
        self._feat_stride = 1. / spatial_scale
        self.pre_nms_topN, self.post_nms_topN, self.nms_thresh, self.min_size = pre_nms_topN, post_nms_topN, nms_thresh, min_size
        self._grid = {}

    def all_anchors(self, H, W, device):
        """(H*W*A, 4) f32 on the device, rows ordered (h, w, a) (:66-88); the f64 -> f32 cast is bbox_transform's
        `boxes.astype(deltas.dtype)` (utils/boxes.py:171)."""
        key = (H, W, str(device))
        if key not in self._grid:
            sx, sy = np.meshgrid(np.arange(0, W) * self._feat_stride, np.arange(0, H) * self._feat_stride, copy=False)
            shifts = np.vstack((sx.ravel(), sy.ravel(), sx.ravel(), sy.ravel())).transpose()
            a = (self._anchors[np.newaxis, :, :] + shifts[:, np.newaxis, :]).reshape((H * W * self._num_anchors, 4))
            self._grid[key] = torch.from_numpy(a.astype(np.float32)).to(device)
        return self._grid[key]

    def proposals_for_one_image(self, im_info, scores_hwa, deltas_hwa):
        """scores_hwa (H,W,A), deltas_hwa (H,W,4A) f32 on the GPU (channels-last: already the (H, W, A) order the
        reference transposes into, :120-131).  Returns device tensors (proposals (R,4), scores (R,))."""
        H, W, A = scores_hwa.shape
        sc = scores_hwa.reshape(-1).float()
        deltas = deltas_hwa.reshape(-1, 4).float()
        # 4./5. descending score, top pre_nms_topN (:135-147); ties by ascending anchor index (the reference's numpy
        # quicksort leaves their order unspecified)
        order = torch.sort(sc, descending=True, stable=True).indices
        if 0 < self.pre_nms_topN < sc.numel():
            order = order[:self.pre_nms_topN]
        anchors = self.all_anchors(H, W, sc.device)[order]
        sc = sc[order]
        # 1./2. decode and clip (:150-155): HIP kernel, f32 as the reference
        props = box_ops.bbox_transform_clip(anchors, deltas[order], (1.0, 1.0, 1.0, 1.0), (float(im_info[0]), float(im_info[1])))
        # 3. _filter_boxes (:176-200), f32 arithmetic
        ms = float(np.float32(self.min_size) * np.float32(im_info[2]))
        ws = props[:, 2] - props[:, 0] + 1
        hs = props[:, 3] - props[:, 1] + 1
        keep = torch.nonzero((ws >= ms) & (hs >= ms) & ((props[:, 0] + ws / 2.) < float(im_info[1])) &
                             ((props[:, 1] + hs / 2.) < float(im_info[0]))).flatten()
        props, sc = props[keep], sc[keep]
        # 6./7. NMS (Cython semantics: >= thresh, ascending indices = descending score here) and the post-NMS cut (:161-171)
        if self.nms_thresh > 0:
            if props.shape[0] > _C.NMS_MAX_BOXES:
                raise RuntimeError("generate_proposals: %d candidates reach NMS with pre_nms_topN = %d (<= 0 keeps every "
                                   "anchor); the device NMS holds %d - set RPN_PRE_NMS_TOP_N (the reference's configs use "
                                   "6000 / 12000)" % (props.shape[0], self.pre_nms_topN, _C.NMS_MAX_BOXES))
            k = box_ops.nms(torch.cat([props, sc[:, None]], 1), self.nms_thresh)
            if self.post_nms_topN > 0:
                k = k[:self.post_nms_topN]
            props, sc = props[k], sc[k]
        return props, sc

    def forward(self, rpn_cls_prob, rpn_bbox_pred, im_info, is_roidb_none=True, channels_last=False):
        """rpn_cls_prob (N,A,H,W), rpn_bbox_pred (N,4A,H,W) (or (N,H,W,A) / (N,H,W,4A) with channels_last), im_info
        (N,3) [h, w, scale] -> (rois (R,5) [batch, x1, y1, x2, y2], roi_probs (R,1)) float32 ndarrays (:90-103)."""
        im_info = np.asarray(im_info.detach().cpu().numpy() if torch.is_tensor(im_info) else im_info, np.float32)
        if not channels_last:
            rpn_cls_prob, rpn_bbox_pred = rpn_cls_prob.permute(0, 2, 3, 1), rpn_bbox_pred.permute(0, 2, 3, 1)
        rois = np.empty((0, 5), dtype=np.float32)
        probs = np.empty((0, 1), dtype=np.float32)
        for i in range(rpn_cls_prob.shape[0]):
            b, s = self.proposals_for_one_image(im_info[i], rpn_cls_prob[i].contiguous(), rpn_bbox_pred[i].contiguous())
            b, s = b.cpu().numpy(), s.cpu().numpy()
            rois = np.append(rois, np.hstack((i * np.ones((b.shape[0], 1), np.float32), b)), axis=0)
            probs = np.append(probs, s[:, None], axis=0)
        return rois, probs


class single_scale_rpn_outputs(nn.Module):
    """rpn_heads.py:39-130 (sigmoid objectness, no FPN), inference."""

    def __init__(self, dim_in, spatial_scale, sizes=(32, 64, 128, 256, 512), aspect_ratios=(0.5, 1, 2), out_dim=None,
                 pre_nms_topN=6000, post_nms_topN=1000, nms_thresh=0.7, min_size=0):
        super().__init__()
        self.dim_in = dim_in
        self.dim_out = dim_in if out_dim is None else out_dim            # RPN.OUT_DIM_AS_IN_DIM (config.py:764)
        anchors = generate_anchors(stride=1. / spatial_scale, sizes=sizes, aspect_ratios=aspect_ratios)
        num_anchors = anchors.shape[0]
        self.RPN_conv = nn.Conv2d(self.dim_in, self.dim_out, 3, 1, 1)
        self.n_score_out = num_anchors
        self.RPN_cls_score = nn.Conv2d(self.dim_out, self.n_score_out, 1, 1, 0)
        self.RPN_bbox_pred = nn.Conv2d(self.dim_out, num_anchors * 4, 1, 1, 0)
        self.RPN_GenerateProposals = GenerateProposalsOp(anchors, spatial_scale, pre_nms_topN, post_nms_topN, nms_thresh, min_size)
        for m in (self.RPN_conv, self.RPN_cls_score, self.RPN_bbox_pred):
            nn.init.normal_(m.weight, std=0.01)
            nn.init.constant_(m.bias, 0)
        self._pk = None

    def _load_from_state_dict(self, *a, **k):
        _PACK_EPOCH[0] += 1
        return super()._load_from_state_dict(*a, **k)

    def _packed(self):
        dev = self.RPN_conv.weight.device
        if self._pk is None or self._pk[0] != _PACK_EPOCH[0] or self._pk[1].device != dev:
            w3 = self.RPN_conv.weight.detach().float().permute(0, 2, 3, 1).reshape(self.dim_out, -1).to(torch.bfloat16).contiguous()
            # [cls | bbox] 1x1 convs as one GEMM; rows padded to a multiple of 8 for 16-byte stores
            wh = torch.cat([self.RPN_cls_score.weight.detach().float().view(self.n_score_out, -1),
                            self.RPN_bbox_pred.weight.detach().float().view(4 * self.n_score_out, -1)], 0)
            bh = torch.cat([self.RPN_cls_score.bias.detach().float(), self.RPN_bbox_pred.bias.detach().float()], 0)
            pad = (-wh.shape[0]) % 8
            wh = F.pad(wh, (0, 0, 0, pad)).to(torch.bfloat16).contiguous()
            bh = F.pad(bh, (0, pad)).contiguous()
            self._pk = (_PACK_EPOCH[0], w3, self.RPN_conv.bias.detach().float().contiguous(), wh, bh)
        return self._pk[1:]

    def heads(self, x):
        """x (N,C,H,W) f32 or channels-last bf16 (N,H,W,C) on the GPU -> (logits (N,H,W,A), deltas (N,H,W,4A)) f32."""
        w3, b3, wh, bh = self._packed()
        if x.dtype != torch.bfloat16:
            x = x.permute(0, 2, 3, 1).to(torch.bfloat16)
        x = x.contiguous().unsqueeze(1)                                                   # one-frame clip
        h = ops.conv3d(x, w3, b3, (1, 3, 3), (1, 1, 1), (0, 1, 1), relu=True)             # :92
        N, _, H, W, _ = h.shape
        o = ops.gemm_nt(h.view(N * H * W, self.dim_out), wh, bias=bh, out_f32=True)       # :94-96
        A = self.n_score_out
        o = o.view(N, H, W, -1)
        return o[..., :A], o[..., A:5 * A]

    def forward(self, x, im_info, roidb=None):
        """rpn_heads.py:84-130, inference: returns rpn_cls_logits / rpn_bbox_pred (NCHW, as the reference), rpn_rois,
        rpn_roi_probs and the alias `rois`."""
        if self.training:
            raise NotImplementedError("trace_amd implements the inference path; call .eval()")
        with torch.no_grad():
            if not x.is_cuda:
                raise RuntimeError("single_scale_rpn_outputs.forward runs on the GPU (libtrace_amd.so); no CPU path")
            logits, deltas = self.heads(x)
            rois, probs = self.RPN_GenerateProposals(torch.sigmoid(logits), deltas, im_info, roidb is None, channels_last=True)
        return {"rpn_cls_logits": logits.permute(0, 3, 1, 2), "rpn_bbox_pred": deltas.permute(0, 3, 1, 2),
                "rpn_rois": rois, "rpn_roi_probs": probs, "rois": rois}
