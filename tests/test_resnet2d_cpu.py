"""2-D conv bodies on CPU: the oracle and the product's torch path against the outputs of the reference's own
ResNet.py (tests/golden/py_resnet2d.npz, tools/make_goldens_r2b.py), and the state-dict layout."""
import json
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = np.load(os.path.join(ROOT, "tests", "golden", "py_resnet2d.npz"))
BODIES = {"r50_dil2": ((3, 4, 6, 3), "ResNet50_conv5_body_dilation2"), "r101_conv5": ((3, 4, 23, 3), "ResNet101_conv5_body")}


def _image():
    g = torch.Generator().manual_seed(int(G["image_seed"]))
    return torch.randn((1, 3, 96, 128), generator=g) * 50


@pytest.mark.parametrize("tag", sorted(BODIES))
def test_body_matches_reference_outputs(tag):
    from oracle.resnet2d_ref import body_forward
    from trace_amd import resnet2d
    from trace_amd.synthetic import fill_resnet2d_state
    counts, ctor = BODIES[tag]
    net = getattr(resnet2d, ctor)().eval()
    keys = json.loads(bytes(G[tag + "_keys"]).decode())
    sd = net.state_dict()
    assert [[k, list(v.shape)] for k, v in sd.items()] == keys            # same names, order and shapes as the reference
    assert [net.spatial_scale, net.dim_out] == G[tag + "_scale_dim"].tolist()
    fill_resnet2d_state(net, seed=3)
    img = _image()
    want = torch.from_numpy(G[tag + "_out8"])
    stages = []
    got_o = body_forward(net.state_dict(), img, counts, stages_out=stages)
    assert list(got_o.shape) == G[tag + "_shape"].tolist()
    scale = float(want.abs().max())
    assert float((got_o[:, ::8] - want).abs().max()) <= 1e-4 * scale
    st = np.asarray([[float(s.mean()), float(s.std())] for s in stages])
    np.testing.assert_allclose(st, G[tag + "_stats"], rtol=1e-4)
    with torch.no_grad():
        got_p = net(img)                                                   # the product's torch path (CPU tensors)
    assert float((got_p[:, ::8] - want).abs().max()) <= 1e-4 * scale
