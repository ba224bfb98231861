#!/usr/bin/env python3
"""How much do the two halves of the I3D backbone gain from running side by side?  Stem + layer1 + layer2 (HBM- /
issue-bound at 8 clips) of one batch on one branch of a captured graph, layer3 + layer4 (MFMA-bound) of another batch on
a second branch, against the same two chains one after the other.  Decides whether a three-deep software pipeline of the
step (backbone split at layer2 | layer3) is worth building.      python tools/overlap_probe.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from trace_amd import ops
    from trace_amd.resnet3d import ResNet3d, i3d_dc5_config
    from trace_amd.synthetic import randomize_model
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    m = ResNet3d(**i3d_dc5_config())
    randomize_model(m, 1)
    calib = torch.randn((1, 3, 16, 112, 112), generator=torch.Generator().manual_seed(1)) * 50
    m.calibrate_bn(calib)
    m.eval()
    m.to(dev)
    clip = (torch.randn((8, 3, 32, 224, 224), generator=torch.Generator().manual_seed(2)) * 50).to(dev)

    def first_half(x):
        x = m._run_stem(x, False)
        for i, name in enumerate(m.res_layers[:2]):
            blocks = list(getattr(m, name))
            for bi, blk in enumerate(blocks):
                x = blk.run(x, pool_t2=(i == 0 and bi == len(blocks) - 1))
        return x

    def second_half(x):
        for name in m.res_layers[2:]:
            for blk in getattr(m, name):
                x = blk.run(x)
        return x

    with torch.no_grad():
        mid = first_half(clip).clone()
        second_half(mid)
        torch.cuda.synchronize()
        side = torch.cuda.Stream(device=dev)

        def serial():
            first_half(clip)
            second_half(mid)

        def forked():
            main = torch.cuda.current_stream(dev)
            first_half(clip[:1, :, :2, :8, :8].contiguous()) if False else None
            side.wait_stream(main)
            with torch.cuda.stream(side), ops.workspace_base(2):
                second_half(mid)
            first_half(clip)
            main.wait_stream(side)

        res = {}
        for name, fn in (("first_half", lambda: first_half(clip)), ("second_half", lambda: second_half(mid)), ("serial", serial),
                         ("forked", forked)):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with ops.capture(g):
                if name == "forked":
                    # a root node on the capture stream before the fork (forking from an empty capture stream crashed
                    # hipStreamEndCapture on ROCm 7.2)
                    torch.zeros((1,), device=dev)
                fn()
            g.replay()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(5):
                s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s0.record()
                for _ in range(5):
                    g.replay()
                e0.record()
                torch.cuda.synchronize()
                best = min(best, s0.elapsed_time(e0) / 5 * 1e3)
            res[name] = best
        res["gain"] = 1.0 - res["forked"] / res["serial"]
        print(json.dumps(res))


if __name__ == "__main__":
    main()
