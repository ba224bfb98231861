#!/usr/bin/env python3
"""Golden vectors of the reference's Cython overlap helpers, from the reference's own code compiled here
(oracle/build_ref.py:build_cython -> oracle/_ref/cython_bbox*.so):

    cython_bbox.bbox_overlaps          Detectron_pytorch/lib/utils/cython_bbox.pyx:32-73
    cython_bbox_rel.bbox_pair_overlaps lib/utils_rel/cython_bbox_rel.pyx:37-77

    python tools/make_goldens_r3.py      -> tests/golden/cy_overlaps.npz (inputs + outputs; data only)

and of the float64 half of `_C.roi_align_forward`'s dtype dispatch (cpu/ROIAlign_cpu.cpp:243), from the reference's
own .cpp compiled by oracle/build_ref.py:  -> tests/golden/roi_align_fwd_f64.npz
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def boxes(rng, n, lo, hi, degenerate=False):
    x1 = rng.uniform(lo, hi, n)
    y1 = rng.uniform(lo, hi, n)
    w = rng.uniform(0 if degenerate else 1, (hi - lo) / 2, n)
    h = rng.uniform(0 if degenerate else 1, (hi - lo) / 2, n)
    return np.stack([x1, y1, x1 + w, y1 + h], 1).astype(np.float32)


def roi_align_f64():
    """Double inputs with entries float32 cannot hold (so a float32 computation would not reproduce the outputs)."""
    import torch
    from oracle.build_ref import build, load_ref
    ref = load_ref() or build()
    if ref is None:
        raise SystemExit("the reference's extension is not built (is /root/reference present?)")
    rng = np.random.default_rng(64)
    feat = rng.standard_normal((2, 6, 14, 14)) * (1.0 + 1e-9 * rng.standard_normal((2, 6, 14, 14)))
    b = boxes(rng, 12, 0, 200).astype(np.float64) + rng.uniform(0, 1e-7, (12, 4))
    rois = np.concatenate([rng.integers(0, 2, (12, 1)).astype(np.float64), b], 1)
    extra = np.array([[0, 50, 50, 50, 50], [1, 100, 120, 90, 100], [0, -40, -40, 30, 30], [1, 200, 200, 400, 400],
                      [0, 0, 0, 223, 223], [1, 300, 300, 500, 500]], np.float64)
    rois = np.concatenate([rois, extra], 0)
    out = {}
    for sr in (0, 2):
        for scale in (1.0 / 16, 0.25):
            out["out_sr%d_s%g" % (sr, scale)] = ref.roi_align_forward(torch.from_numpy(feat), torch.from_numpy(rois),
                                                                      scale, 7, 7, sr).numpy()
    out["out_5x3_sr0"] = ref.roi_align_forward(torch.from_numpy(feat), torch.from_numpy(rois), 1.0 / 16, 5, 3, 0).numpy()
    assert all(v.dtype == np.float64 for v in out.values())
    dst = os.path.join(ROOT, "tests", "golden", "roi_align_fwd_f64.npz")
    np.savez_compressed(dst, feat=feat, rois=rois, **out)
    print("wrote", dst)


def main():
    roi_align_f64()
    from oracle import build_ref
    build_ref.build_cython()
    cb, cbr = build_ref.load_cython("cython_bbox"), build_ref.load_cython("cython_bbox_rel")
    if cb is None or cbr is None:
        raise SystemExit("the reference's Cython modules are not built (is /root/reference present?)")
    rng = np.random.default_rng(2026)
    out = {}
    cases = {"rand": (boxes(rng, 37, 0, 224), boxes(rng, 53, 0, 224)),
             "dense": (boxes(rng, 64, 100, 140), boxes(rng, 64, 100, 140)),          # heavy overlap
             "touch": (np.array([[0, 0, 9, 9], [10, 10, 19, 19], [0, 0, 0, 0], [5, 5, 5, 5]], np.float32),
                       np.array([[9, 9, 20, 20], [0, 0, 9, 9], [0, 0, 0, 0], [4, 4, 6, 6], [30, 30, 40, 40]], np.float32)),
             "big": (boxes(rng, 300, 0, 1333), boxes(rng, 200, 0, 1333))}
    for k, (a, b) in cases.items():
        out["ov_%s_a" % k], out["ov_%s_b" % k] = a, b
        out["ov_%s_out" % k] = np.asarray(cb.bbox_overlaps(a, b), np.float32)
    pcases = {"rand": (boxes(rng, 380, 0, 224), boxes(rng, 380, 0, 224)),
              "dense": (boxes(rng, 100, 50, 90), boxes(rng, 100, 50, 90)),
              "touch": (cases["touch"][0], cases["touch"][1][:4])}
    for k, (a, b) in pcases.items():
        out["po_%s_a" % k], out["po_%s_b" % k] = a, b
        out["po_%s_out" % k] = np.asarray(cbr.bbox_pair_overlaps(a, b), np.float32)
    path = os.path.join(ROOT, "tests", "golden", "cy_overlaps.npz")
    np.savez_compressed(path, **out)
    print(path, {k: v.shape for k, v in out.items() if k.endswith("_out")})


if __name__ == "__main__":
    main()
