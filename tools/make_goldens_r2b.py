"""Fixture for the 2-D conv bodies, produced by RUNNING THE REFERENCE'S OWN ResNet.py in the build container
(environment: tools/ref_env.py); data only.

    python tools/make_goldens_r2b.py

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


def main():
    cfg = ref_env.setup()
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
