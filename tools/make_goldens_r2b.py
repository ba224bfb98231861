"""Fixture for the 2-D conv bodies, produced by RUNNING THE REFERENCE'S OWN ResNet.py in the build container
(environment: tools/ref_env.py); data only.

    python tools/make_goldens_r2b.py

  py_rpn.npz        `single_scale_rpn_outputs` + `GenerateProposalsOp` (rpn_heads.py:39-130, generate_proposals.py:12-171)
                    at 64 input channels on a seeded 20 x 30 map, VidVRD config (5 sizes x 3 ratios, 6000 -> NMS 0.7 ->
                    1000): logits, deltas, anchors, proposals and their probabilities.  Weights seeded per key.
  py_resnet2d.npz   Detectron_pytorch/lib/modeling/ResNet.py `ResNet50_conv5_body_dilation2()` (TRACE's Prd_RCNN2D) and
                    `ResNet101_conv5_body()` (Conv_Body) under the VidVRD config's RESNETS.STRIDE_1X1 False /
                    RES5_DILATION 2, weights from trace_amd.synthetic.fill_resnet2d_state (seeded per state-dict key,
                    so the fixture holds no weights), run on one seeded 96 x 128 image.  Stored per body: the
                    state-dict keys and shapes, spatial_scale, dim_out, every 8th channel of the output, and the
                    mean / std of each stage's output.
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_env  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def rpn(cfg):
    """single_scale_rpn_outputs (rpn_heads.py:39-130) + GenerateProposalsOp (generate_proposals.py:12-171) on one
    seeded feature map.  The Cython NMS (utils/cython_nms.pyx) cannot be built here; `utils.boxes.nms` is pointed at
    the reference's own C++ implementation of the same function (csrc/cpu/nms_cpu.cpp through model._C, built by
    oracle/build_ref.py) for this one call, as the docstring of box_utils.nms allows (dets (N,5) -> ascending keep)."""
    import zlib
    import modeling.rpn_heads as RH
    import utils.boxes as box_utils
    from model import _C as ref_C
    cfg.RPN.SIZES = (32, 64, 128, 256, 512)          # configs/vidvrd/...dc5_2d_new.yaml:52-53
    cfg.TEST.RPN_PRE_NMS_TOP_N, cfg.TEST.RPN_POST_NMS_TOP_N = 6000, 1000      # :68-69
    cfg.MODEL.FASTER_RCNN = True

    def ref_nms(dets, thresh):
        d = torch.from_numpy(np.ascontiguousarray(dets[:, :4], np.float32))
        sc = torch.from_numpy(np.ascontiguousarray(dets[:, 4], np.float32))
        return ref_C.nms(d, sc, float(thresh)).numpy()
    box_utils.nms = ref_nms
    net = RH.single_scale_rpn_outputs(64, 1.0 / 16)
    net.eval()
    sd = {}
    for k, v in net.state_dict().items():
        g = torch.Generator().manual_seed(zlib.crc32(k.encode()) & 0x7fffffff)
        std = {"RPN_conv.weight": 0.05, "RPN_cls_score.weight": 0.3, "RPN_bbox_pred.weight": 0.05}.get(k, 0.1)
        sd[k] = torch.randn(v.shape, generator=g) * std
    net.load_state_dict(sd)
    g = torch.Generator().manual_seed(33)
    x = torch.randn((1, 64, 20, 30), generator=g)
    im_info = torch.tensor([[320.0, 480.0, 1.6]])
    with torch.no_grad():
        ret = net(x, im_info, None)
    out = dict(x_seed=np.int64(33), im_info=im_info.numpy(), logits=ret["rpn_cls_logits"].numpy(), bbox_pred=ret["rpn_bbox_pred"].numpy(),
               rois=np.asarray(ret["rpn_rois"], np.float32), probs=np.asarray(ret["rpn_roi_probs"], np.float32),
               anchors=np.asarray(net.RPN_GenerateProposals._anchors, np.float64),
               cfg=np.frombuffer(json.dumps(dict(sizes=list(cfg.RPN.SIZES), ratios=list(cfg.RPN.ASPECT_RATIOS), pre=6000, post=1000,
                                                 nms=float(cfg.TEST.RPN_NMS_THRESH), min_size=float(cfg.TEST.RPN_MIN_SIZE))).encode(), dtype=np.uint8),
               keys=np.frombuffer(json.dumps([[k, list(v.shape)] for k, v in net.state_dict().items()]).encode(), dtype=np.uint8))
    assert ret["rois"] is ret["rpn_rois"]
    np.savez_compressed(os.path.join(OUT, "py_rpn.npz"), **out)
    print("rpn: %d proposals from %d anchors, top prob %.4f" % (out["rois"].shape[0], 20 * 30 * out["anchors"].shape[0], float(out["probs"].max())))


def main():
    cfg = ref_env.setup()
    rpn(cfg)
    cfg.RESNETS.STRIDE_1X1 = False      # configs/vidvrd/vidvrd_res101xi3d50_gt_boxes_dc5_2d_new.yaml:41
    cfg.RESNETS.RES5_DILATION = 2       # :45
    import modeling.ResNet as R
    from trace_amd.synthetic import fill_resnet2d_state
    g = torch.Generator().manual_seed(21)
    img = torch.randn((1, 3, 96, 128), generator=g) * 50
    out = {"image_seed": np.int64(21)}
    for tag, ctor in (("r50_dil2", R.ResNet50_conv5_body_dilation2), ("r101_conv5", R.ResNet101_conv5_body)):
        net = ctor()
        net.eval()
        fill_resnet2d_state(net, seed=3)
        x = img
        stats = []
        with torch.no_grad():
            for i in range(net.convX):
                x = getattr(net, "res%d" % (i + 1))(x)
                stats.append([float(x.mean()), float(x.std())])
        sd = net.state_dict()
        out[tag + "_keys"] = np.frombuffer(json.dumps([[k, list(v.shape)] for k, v in sd.items()]).encode(), dtype=np.uint8)
        out[tag + "_out8"] = x[:, ::8].numpy().astype(np.float32)
        out[tag + "_shape"] = np.asarray(x.shape, np.int64)
        out[tag + "_stats"] = np.asarray(stats, np.float64)
        out[tag + "_scale_dim"] = np.asarray([net.spatial_scale, net.dim_out], np.float64)
        print(tag, tuple(x.shape), stats[-1], len(sd), "keys")
    np.savez_compressed(os.path.join(OUT, "py_resnet2d.npz"), **out)
    print("py_resnet2d.npz %d bytes" % os.path.getsize(os.path.join(OUT, "py_resnet2d.npz")))


if __name__ == "__main__":
    main()
