#!/usr/bin/env python3
"""Bandwidth-bound kernels of the path against the HBM roofline, each timed alone with HIP events
around a captured graph of `reps` launches (the host's per-call cost stays out of the bracket):
ROIAlign (object-head call of the bench workload; adaptive-grid region / temporal calls; the fused avg variant),
pair pooling / combine, NMS (6 000 RPN boxes @0.7; 35 classes x <= 300 boxes @0.5 grouped).
Prints one JSON object; bench.py imports `measure()` for its roofline_* entries.

    python tools/bench_kernels.py            (TRACE_ROI_ALG=1|2 selects the older ROIAlign kernels for A/B)
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HBM_PEAK = 8.0e12


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    from trace_amd import ops
    g = torch.cuda.CUDAGraph()
    with ops.capture(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record()
        g.replay()
        e0.record()
        torch.cuda.synchronize()
        best = min(best, s0.elapsed_time(e0) / reps * 1e3)
    return best


def entry(kernel, us, nbytes, **extra):
    d = dict(bound="hbm", achieved=nbytes / us / 1e3, peak=HBM_PEAK / 1e9, unit="GB/s", frac=nbytes / (us * 1e-6) / HBM_PEAK,
             kernel=kernel, avg_launch_us=us, algorithmic_bytes_per_launch=int(nbytes))
    d.update(extra)
    return d


def measure(dev, B=8, n=20, Tn=4, C=2048, fm=14, seed=1, nms=True):
    from trace_amd import _C, ops
    from trace_amd import boxes as tb
    from trace_amd.synthetic import synth_boxes
    rng = np.random.default_rng(seed)
    out = {}
    blob = torch.randn((B, fm, fm, C), device=dev).to(torch.bfloat16)
    rois = np.concatenate([np.concatenate([np.full((n, 1), b, np.float32), synth_boxes(rng, n, 2.0, 16.0 * fm)], 1)
                           for b in range(B)], 0)
    r = torch.from_numpy(rois).to(dev)
    K = rois.shape[0]
    nb = K * 49 * C * 2 + blob.numel() * 2 + K * 20
    out["roi_align"] = entry("trace::roi_align_fwd_mfma_nhwc_bf16_kernel", timed(lambda: ops.roi_align_nhwc(blob, r, 1 / 16., 7, 2)), nb,
                             rois=K, channels=C, call="object heads: %d rois x %d ch, 7x7, sampling_ratio 2" % (K, C))
    out["roi_align_adaptive"] = entry("trace::roi_align_fwd_mfma_nhwc_bf16_kernel", timed(lambda: ops.roi_align_nhwc(blob, r, 1 / 16., 7, 0)), nb,
                                      rois=K, channels=C, call="region rois: adaptive grid (sampling_ratio 0)")
    tm = torch.randn((B * Tn, fm, fm, C), device=dev).to(torch.bfloat16)
    tr = np.concatenate([np.concatenate([np.full((n, 1), f, np.float32), rois[(f // Tn) * n:(f // Tn + 1) * n, 1:]], 1)
                         for f in range(B * Tn)], 0)
    trd = torch.from_numpy(tr).to(dev)
    Kt = tr.shape[0]
    out["roi_align_avg"] = entry("trace::roi_align_avg_nhwc_bf16_kernel", timed(lambda: ops.roi_align_avg_nhwc(tm, trd, 1 / 16., 7, 0)),
                                 Kt * C * 2 + tm.numel() * 2 + Kt * 20, rois=Kt, channels=C,
                                 call="temporal object rois: ROIAlign + 7x7 mean, %d rois" % Kt)
    # pair kernels (model_builder_rel.py:1524-1612 fused): union-box pooling of [V | W_A V | W_B V] and the
    # attention-weighted combine, B * n(n-1) pairs, swapped pairs pooled once
    from trace_amd.config import get_cfg
    from trace_amd.model_builder_rel import swapped_pair_index
    cfg = get_cfg()
    D = int(cfg.LOW_FEAT_DIM)
    X = np.concatenate([np.repeat(np.arange(n), n - 1) + b * n for b in range(B)])
    Y = np.concatenate([np.concatenate([np.delete(np.arange(n), i) for i in range(n)]) + b * n for b in range(B)])
    P = X.shape[0]
    ob = rois[:, 1:]
    un = np.concatenate([rois[X, :1], np.minimum(ob[X, :2], ob[Y, :2]), np.maximum(ob[X, 2:], ob[Y, 2:])], 1).astype(np.float32)
    und = torch.from_numpy(un).to(dev)
    part = torch.from_numpy(swapped_pair_index(X, Y)).to(dev)
    Xd, Yd = torch.from_numpy(X.astype(np.int32)).to(dev), torch.from_numpy(Y.astype(np.int32)).to(dev)
    vab = torch.randn((B, fm, fm, 3 * D), device=dev).to(torch.bfloat16)
    qk = torch.randn((2 * B * n, B * fm * fm), device=dev)
    bias = torch.randn((D,), device=dev)
    uniq = int((part.cpu().numpy() < 0).sum() + (part.cpu().numpy() > np.arange(P)).sum())
    pool_bytes = uniq * 49 * 3 * D * 2 + vab.numel() * 2 + P * 24
    out["pair_pool"] = entry("trace::pair_pool_kernel", timed(lambda: ops.pair_pool(vab, und, 1 / 16., 7, partner=part)),
                             pool_bytes, pairs=P, pooled_rows=uniq, D=D,
                             call="union-box pooling of %d pairs (%d unique boxes) x 3 x %d ch" % (P, uniq, D))
    pooled = ops.pair_pool(vab, und, 1 / 16., 7, partner=part)
    comb_bytes = uniq * 49 * 3 * D * 2 + P * 49 * D * 2 + P * 49 * 2 * 4 * 4
    out["pair_combine"] = entry("trace::pair_combine_kernel", timed(lambda: ops.pair_combine(pooled, (B, fm, fm), qk, und, Xd, Yd, bias, 1 / 16., 7, partner=part)),
                                comb_bytes, pairs=P, D=D, call="softmax-weighted combine of the pooled maps -> (P,49,%d)" % D)
    fused_bytes = P * 49 * D * 2 + vab.numel() * 2 + P * 49 * 2 * 4 * 4
    out["pair_infer"] = entry("trace::pair_infer_kernel", timed(lambda: ops.pair_infer(vab, qk, und, Xd, Yd, bias, 1 / 16., 7, partner=part)),
                              fused_bytes, pairs=P, D=D, call="pool + combine in one launch (no pooled map in HBM)")
    vabT = torch.zeros((B, 3 * D, 224), dtype=torch.bfloat16, device=dev)
    vabT[:, :, :fm * fm] = vab.view(B, fm * fm, 3 * D).permute(0, 2, 1)
    vbias = torch.randn((3 * D,), device=dev)
    work = torch.from_numpy(ops.pair_work_records(X, Y, un, part.cpu().numpy())).to(dev)
    out["pair_mfma"] = entry("trace::pair_mfma_kernel", timed(lambda: ops.pair_infer_mfma(vabT, (fm, fm), vbias, qk, und, Xd, Yd, bias, 1 / 16., 7, partner=part, work=work)),
                             fused_bytes, pairs=P, D=D, call="pool (MFMA) + combine in one launch, channel-major maps")
    if not nms:
        return out
    # NMS: SURVEY 8d stress sizes.  Algorithmic bytes = N*20 in + N*8 out (compute-light: boxes/s is the figure)
    N = 6000
    bx = torch.from_numpy(synth_boxes(rng, N, 3.0, 800.0)).to(dev)
    sc = torch.from_numpy((rng.permutation(N) / N).astype(np.float32)).to(dev)
    lib_ws = torch.empty((int(__import__("trace_amd._lib", fromlist=["lib"]).lib().trace_nms_workspace_bytes(N)),), dtype=torch.uint8, device=dev)
    from trace_amd import _lib as L
    keep = torch.empty((N,), dtype=torch.int64, device=dev)
    num = torch.zeros((1,), dtype=torch.int32, device=dev)

    def nms_call():
        L.check(L.lib().trace_nms(L.ptr(bx), L.ptr(sc), N, 0.7, 0, L.ptr(keep), L.ptr(num), L.ptr(lib_ws), lib_ws.numel(), L.stream()))
    us = timed(nms_call, reps=5)
    out["nms_rpn"] = entry("trace::nms_{sort,mask,reduce,compact}_kernel", us, N * 28, boxes=N, thresh=0.7,
                           boxes_per_s=N / (us * 1e-6), kept=int(num.item()), call="6000 RPN boxes @0.7 (generate_proposals.py:164)")
    per = [int(v) for v in rng.integers(1, 301, 35)]
    cls = np.concatenate([np.full(k, c + 1, np.int32) for c, k in enumerate(per)])
    Ng = cls.shape[0]
    bxg = torch.from_numpy(synth_boxes(rng, Ng, 3.0, 800.0)).to(dev)
    scg = torch.from_numpy((rng.permutation(Ng) / Ng).astype(np.float32)).to(dev)
    clg = torch.from_numpy(cls).to(dev)
    wsg = torch.empty((int(L.lib().trace_nms_workspace_bytes(Ng)),), dtype=torch.uint8, device=dev)
    keepg = torch.empty((Ng,), dtype=torch.int64, device=dev)

    def nmsg_call():
        L.check(L.lib().trace_nms_grouped(L.ptr(bxg), L.ptr(scg), L.ptr(clg), Ng, 0.5, 0, L.ptr(keepg), L.ptr(num), L.ptr(wsg),
                                          wsg.numel(), L.stream()))
    us = timed(nmsg_call, reps=5)
    out["nms_per_class"] = entry("trace::nms_{sort,mask,reduce,compact}_kernel", us, Ng * 32, boxes=Ng, classes=35, thresh=0.5,
                                 boxes_per_s=Ng / (us * 1e-6), call="35 classes x <=300 boxes @0.5 in one grouped pass")
    return out


if __name__ == "__main__":
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    res = measure(dev)
    print(json.dumps(res, indent=1))
