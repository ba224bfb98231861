"""Generate tests/golden/*.npz from the REFERENCE ITSELF (runs only in the build
container, where /root/reference exists).

Source of truth: oracle/_ref/trace_ref_C.so = the reference's own
csrc/{vision.cpp,cpu/ROIAlign_cpu.cpp,cpu/nms_cpu.cpp} compiled unmodified by
oracle/build_ref.py.  Fixtures hold inputs + the reference's outputs only.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.build_ref import build, load_ref  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def synth_boxes(rng, n, k=2.0, size=224.0):
    """x1,y1 ~ U(0,160), w,h ~ U(16,64k) clipped to size-1."""
    x1 = rng.uniform(0, 160, n)
    y1 = rng.uniform(0, 160, n)
    w = rng.uniform(16, 64 * k, n)
    h = rng.uniform(16, 64 * k, n)
    return np.stack([x1, y1, np.minimum(x1 + w, size - 1), np.minimum(y1 + h, size - 1)], 1).astype(np.float32)


def main():
    ref = load_ref() or build()
    assert ref is not None, "reference sources not available"
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(1)

    # ---- ROIAlign forward: sr=0 (adaptive) and sr=2, incl. degenerate / OOB rois
    feat = rng.standard_normal((2, 16, 14, 14)).astype(np.float32)
    boxes = synth_boxes(rng, 10)
    rois = np.concatenate([rng.integers(0, 2, (10, 1)).astype(np.float32), boxes], 1)
    extra = np.array([
        [0, 50, 50, 50, 50],          # zero-size -> forced 1x1 (ROIAlign_cpu.cpp:156-157)
        [1, 100, 120, 90, 100],       # inverted -> forced 1x1
        [0, -40, -40, 30, 30],        # partly outside top-left
        [1, 200, 200, 400, 400],      # beyond bottom-right (samples > H -> 0)
        [0, 0, 0, 223, 223],          # whole image
        [1, 300, 300, 500, 500],      # fully outside
    ], np.float32)
    rois = np.concatenate([rois, extra], 0).astype(np.float32)
    cases = {}
    for sr in (0, 2):
        for scale in (1.0 / 16, 0.25):
            out = ref.roi_align_forward(torch.from_numpy(feat), torch.from_numpy(rois), scale, 7, 7, sr).numpy()
            cases["out_sr%d_s%g" % (sr, scale)] = out
    out53 = ref.roi_align_forward(torch.from_numpy(feat), torch.from_numpy(rois), 1.0 / 16, 5, 3, 0).numpy()
    np.savez_compressed(os.path.join(OUT, "roi_align_fwd.npz"), feat=feat, rois=rois, out_5x3_sr0=out53, **cases)

    # ---- NMS: distinct scores, several sizes / thresholds (nms_cpu.cpp == cython_nms.pyx algorithm)
    nms_cases = {}
    for name, n, k, thr in (("n10", 10, 2.0, 0.5), ("n50", 50, 2.0, 0.5), ("n300", 300, 1.5, 0.5),
                            ("n1000_rpn", 1000, 2.0, 0.7), ("n130_t03", 130, 3.0, 0.3)):
        b = synth_boxes(rng, n, k)
        s = rng.permutation(n).astype(np.float32) / n * 0.95 + 0.05  # distinct
        keep = ref.nms(torch.from_numpy(b), torch.from_numpy(s.astype(np.float32)), thr).numpy()
        nms_cases[name + "_boxes"] = b
        nms_cases[name + "_scores"] = s.astype(np.float32)
        nms_cases[name + "_thresh"] = np.float32(thr)
        nms_cases[name + "_keep"] = keep.astype(np.int64)
    # exact-threshold case: IoU == thresh must be suppressed by the CPU path (>=)
    b = np.array([[0, 0, 9, 9], [0, 0, 9, 4], [20, 20, 29, 29]], np.float32)  # IoU(0,1) = 50/100 = 0.5
    s = np.array([0.9, 0.8, 0.7], np.float32)
    keep = ref.nms(torch.from_numpy(b), torch.from_numpy(s), 0.5).numpy()
    nms_cases["eq_boxes"], nms_cases["eq_scores"], nms_cases["eq_thresh"], nms_cases["eq_keep"] = b, s, np.float32(0.5), keep.astype(np.int64)
    np.savez_compressed(os.path.join(OUT, "nms.npz"), **nms_cases)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
