#!/usr/bin/env python3
"""The 2-D conv bodies' own bench line: Conv_Body = ResNet-101 (res5 dilated) on an 800 x 1333 key
frame and Prd_RCNN2D = ResNet-50-dil2 on the relation frames of 8 clips (224^2), bf16 implicit-GEMM convs with the
frozen BN folded, replayed from a HIP graph.  Prints one JSON line per body: ms per forward, algorithmic TFLOP/s
(conv MACs x 2, from the per-launch trace), fraction of the dense bf16 MFMA peak.

    python tools/bench_bodies.py [--steps 20]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
MFMA_BF16_PEAK = 2.5e15


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    args = ap.parse_args()
    from trace_amd import ops, resnet2d
    from trace_amd.synthetic import fill_resnet2d_state
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    for name, ctor, shape in (("Conv_Body R101-DC5, 800x1333 key frame", resnet2d.ResNet101_conv5_body, (1, 3, 800, 1333)),
                              ("Prd_RCNN2D R50-dil2, 8 x 224^2 frames", resnet2d.ResNet50_conv5_body_dilation2, (8, 3, 224, 224))):
        net = fill_resnet2d_state(ctor().eval(), 1).to(dev)
        x = torch.randn(shape, device=dev) * 50
        for _ in range(2):
            net.forward_nhwc(x)
        tr = []
        ops.IGEMM_TRACE = tr
        net.forward_nhwc(x)
        torch.cuda.synchronize()
        ops.IGEMM_TRACE = None
        flops = sum(t[2] for t in tr)
        g = torch.cuda.CUDAGraph()
        with ops.capture(g):
            out = net.forward_nhwc(x)
        g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            g.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        print(json.dumps({"body": name, "input": list(shape), "output": list(out.shape), "ms": dt * 1e3, "gflop": flops / 1e9,
                          "tflops": flops / dt / 1e12, "frac_mfma_peak": flops / dt / MFMA_BF16_PEAK,
                          "conv_launches": len(tr), "frames_per_sec": shape[0] / dt}))


if __name__ == "__main__":
    main()
