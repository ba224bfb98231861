"""RPN proposal oracle against the outputs of the reference's own rpn_heads.py / generate_proposals.py
(tests/golden/py_rpn.npz, tools/make_goldens_r2b.py)."""
import json
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = np.load(os.path.join(ROOT, "tests", "golden", "py_rpn.npz"))
CFG = json.loads(bytes(G["cfg"]).decode())


def test_anchors_match_reference():
    from oracle.np_oracle import generate_anchors
    from trace_amd.rpn import generate_anchors as prod_anchors
    a = generate_anchors(16, CFG["sizes"], CFG["ratios"])
    assert a.dtype == np.float64 and np.array_equal(a, G["anchors"])
    assert np.array_equal(prod_anchors(16, CFG["sizes"], CFG["ratios"]), G["anchors"])


def test_generate_proposals_oracle_matches_reference():
    from oracle import c_oracle
    from oracle.np_oracle import generate_proposals
    prob = torch.sigmoid(torch.from_numpy(G["logits"])).numpy()
    props, sc, pos = generate_proposals(prob[0], G["bbox_pred"][0], G["im_info"][0], G["anchors"], 16.0, CFG["pre"], CFG["post"],
                                        CFG["nms"], CFG["min_size"], nms_fn=lambda b, s, t: c_oracle.nms(b, s, t, 0))
    assert props.shape[0] == G["rois"].shape[0] == 1000
    assert np.array_equal(props, G["rois"][:, 1:]) and np.array_equal(sc, G["probs"]) and np.all(G["rois"][:, 0] == 0)
    assert np.all(np.diff(sc[:, 0]) <= 0)


def test_module_state_dict_layout():
    from trace_amd.rpn import single_scale_rpn_outputs
    net = single_scale_rpn_outputs(64, 1.0 / 16)
    assert [[k, list(v.shape)] for k, v in net.state_dict().items()] == json.loads(bytes(G["keys"]).decode())
