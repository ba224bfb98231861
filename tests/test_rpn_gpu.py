"""RPN heads + proposal generation on the GPU against the reference's own outputs (tests/golden/py_rpn.npz) and the
fp32 torch convs."""
import json
import os
import zlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = np.load(os.path.join(ROOT, "tests", "golden", "py_rpn.npz"))
CFG = json.loads(bytes(G["cfg"]).decode())


def seeded_rpn(dim_in=64):
    from trace_amd.rpn import single_scale_rpn_outputs
    net = single_scale_rpn_outputs(dim_in, 1.0 / 16, sizes=CFG["sizes"], aspect_ratios=CFG["ratios"], pre_nms_topN=CFG["pre"],
                                   post_nms_topN=CFG["post"], nms_thresh=CFG["nms"], min_size=CFG["min_size"]).eval()
    sd = {}
    for k, v in net.state_dict().items():                       # the generator's seeding (tools/make_goldens_r2b.py)
        g = torch.Generator().manual_seed(zlib.crc32(k.encode()) & 0x7fffffff)
        std = {"RPN_conv.weight": 0.05, "RPN_cls_score.weight": 0.3, "RPN_bbox_pred.weight": 0.05}.get(k, 0.1)
        sd[k] = torch.randn(v.shape, generator=g) * std
    net.load_state_dict(sd)
    return net


def test_generate_proposals_on_reference_maps_is_exact():
    """The device op on the REFERENCE's logits / deltas: which anchors survive (top-6000 cut, min-size filter, NMS 0.7,
    top-1000) and their scores bit-exact, boxes to 1e-5 of the image size (expf vs numpy's exp)."""
    net = seeded_rpn().to(DEV)
    prob = torch.sigmoid(torch.from_numpy(G["logits"])).to(DEV)
    rois, probs = net.RPN_GenerateProposals(prob, torch.from_numpy(G["bbox_pred"]).to(DEV), torch.from_numpy(G["im_info"]))
    assert rois.dtype == np.float32 and rois.shape == G["rois"].shape and probs.shape == G["probs"].shape
    assert np.array_equal(probs, G["probs"])                                 # the same anchors in the same order
    np.testing.assert_allclose(rois, G["rois"], rtol=0, atol=480 * 1e-5)
    # channels-last inputs take the same path
    r2, p2 = net.RPN_GenerateProposals(prob.permute(0, 2, 3, 1).contiguous(), torch.from_numpy(G["bbox_pred"]).to(DEV).permute(0, 2, 3, 1).contiguous(),
                                       G["im_info"], channels_last=True)
    assert np.array_equal(r2, rois) and np.array_equal(p2, probs)


def test_rpn_heads_match_fp32_convs_and_reference_logits():
    net = seeded_rpn()
    g = torch.Generator().manual_seed(int(G["x_seed"]))
    x = torch.randn((1, 64, 20, 30), generator=g)
    sd = net.state_dict()
    conv = F.relu(F.conv2d(x, sd["RPN_conv.weight"], sd["RPN_conv.bias"], padding=1))
    ref_l = F.conv2d(conv, sd["RPN_cls_score.weight"], sd["RPN_cls_score.bias"])
    ref_d = F.conv2d(conv, sd["RPN_bbox_pred.weight"], sd["RPN_bbox_pred.bias"])
    np.testing.assert_allclose(ref_l.numpy(), G["logits"], atol=2e-5)        # the fp32 restatement is the reference's map
    np.testing.assert_allclose(ref_d.numpy(), G["bbox_pred"], atol=2e-5)
    net.to(DEV)
    ret = net(x.to(DEV), torch.from_numpy(G["im_info"]))
    for got, ref in ((ret["rpn_cls_logits"], ref_l), (ret["rpn_bbox_pred"], ref_d)):
        assert got.shape == ref.shape
        e = float((got.cpu() - ref).norm() / ref.norm())
        assert e < 2e-2, e
    assert ret["rois"] is ret["rpn_rois"] and ret["rpn_rois"].shape[1] == 5 and ret["rpn_rois"].shape[0] <= CFG["post"]
    # proposals from the bf16 maps: the same op on slightly different scores - most of the reference's boxes are found
    from oracle import c_oracle
    iou = c_oracle.bbox_overlaps(G["rois"][:200, 1:], ret["rpn_rois"][:, 1:])
    assert float((iou.max(1) > 0.9).mean()) > 0.9


def test_rpn_drives_the_detected_box_flow():
    """Generalized_RCNN.forward without gt / pre-computed boxes: RPN -> Box_Head / Box_Outs -> per-class NMS
    (model_builder_rel.py:733, :908-935) -> relation head, end to end on the device."""
    from trace_amd.model_builder_rel import Generalized_RCNN
    from trace_amd.rpn import single_scale_rpn_outputs
    from trace_amd.synthetic import fill_resnet2d_state, make_inputs, randomize_model
    m = Generalized_RCNN()
    randomize_model(m, 1)
    fill_resnet2d_state(m.Conv_Body, 2)
    fill_resnet2d_state(m.Prd_RCNN2D, 3)
    m.RPN = single_scale_rpn_outputs(m.Conv_Body.dim_out, m.Conv_Body.spatial_scale, post_nms_topN=300)
    assert "RPN.RPN_conv.weight" in m.state_dict()
    m.eval().to(DEV)
    frames, _, _, _ = make_inputs(1, 8, 4, 224, seed=3)
    frames = frames.to(DEV)
    ret = m(frames[:, 4].contiguous(), frames, torch.tensor([[224.0, 224.0, 1.0]]), dataset_name="ag_val", bf_cur_len=4)
    n = ret["det_rois"].shape[0]
    assert 1 <= n <= m.cfg.DETECTIONS_PER_IM and ret["prd_scores"].shape[0] == ret["sbj_rois"].shape[0] > 0
    assert torch.isfinite(ret["prd_scores"]).all()
