#!/usr/bin/env python3
"""Per-launch table of the conv / GEMM family over one step of the bench workload (eager launches bracketed by events
on the launch stream, ops.IGEMM_TRACE): shape, time, TFLOP/s, algorithmic GB/s.  Diagnostic for kernel work; the
bench's roofline object is the aggregate of the same events.

    python tools/perf_layers.py [--batch 8] [--frames 32] [--reps 5] [--backbone-only]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--objects", type=int, default=20)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--backbone-only", action="store_true")
    ap.add_argument("--u8", action="store_true", help="feed decoded uint8 frames (B,T,H,W,3) instead of the f32 clip")
    args = ap.parse_args()
    from trace_amd import ops
    from trace_amd.model_builder_rel import Generalized_RCNN
    from trace_amd.synthetic import make_inputs, randomize_model
    dev = torch.device("cuda", 0)
    torch.manual_seed(1)
    m = Generalized_RCNN()
    randomize_model(m, 1)
    m.eval().to(dev)
    frames, blob, feat, rois = make_inputs(args.batch, args.frames, args.objects, 224, seed=1)
    frames, blob, feat = frames.to(dev), blob.to(dev), feat.to(dev)
    if args.u8:
        frames = torch.randint(0, 256, (args.batch, args.frames, 224, 224, 3), dtype=torch.uint8, device=dev)
        m.Prd_RCNN.pixel_means = [102.9801, 115.9465, 122.7717]

    def step():
        if args.backbone_only:
            m.Prd_RCNN.forward_ndhwc(frames, in_ntchw=True)
        else:
            m.forward_clips(frames, blob, feat, rois, topk=100, prd_k=1)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    runs = []
    for _ in range(args.reps):
        tr = []
        ops.IGEMM_TRACE = tr
        step()
        torch.cuda.synchronize()
        ops.IGEMM_TRACE = None
        runs.append(tr)
    empty = []
    for _ in range(200):
        s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record()
        e0.record()
        empty.append((s0, e0))
    torch.cuda.synchronize()
    ov = float(np.median([a.elapsed_time(b) for a, b in empty]))
    n = len(runs[0])
    tot_us = tot_fl = 0.0
    print("%3s %9s %8s %8s  %s" % ("#", "us", "TFLOP/s", "GB/s", "launch"))
    for i in range(n):
        us = float(np.median([max(r[i][0].elapsed_time(r[i][1]) - ov, 1e-3) for r in runs])) * 1e3
        fl, nb, lab = runs[0][i][2], runs[0][i][3], runs[0][i][4]
        tot_us += us
        tot_fl += fl
        print("%3d %9.1f %8.1f %8.0f  %s" % (i, us, fl / us / 1e6, nb / us / 1e3, lab))
    print("total %.1f us, %.1f GFLOP, %.1f TFLOP/s (event bracket overhead %.1f us removed per launch)" % (
        tot_us, tot_fl / 1e9, tot_fl / tot_us / 1e6, ov * 1e3))


if __name__ == "__main__":
    main()
