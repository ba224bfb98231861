"""Host-side logic of the product that needs no GPU, against the reference-generated goldens."""
import json

import numpy as np

from conftest import synth_boxes


def jload(a):
    return json.loads(bytes(a.tobytes()).decode())


def test_relpn_pair_indices_match_reference(golden):
    from trace_amd.boxes import relpn_pair_indices
    g = golden("py_relpn.npz")
    for key in jload(g["cases"]):
        n = g[key + "_det_rois"].shape[0]
        s, o = relpn_pair_indices(n, g[key + "_labels"], jload(g[key + "_dataset"]))
        assert np.array_equal(s, g[key + "_out_sbj_inds"]) and np.array_equal(o, g[key + "_out_obj_inds"]), key


def test_ovlp_filter_equals_pair_overlap_gt_zero():
    """relpn_heads.py:94-105: keep pairs with IoU > 0 ('+1' areas, cython_bbox_rel.pyx:37-77)."""
    from trace_amd.boxes import ovlp_filter
    rng = np.random.default_rng(2)
    a, b = synth_boxes(rng, 200, 0.5), synth_boxes(rng, 200, 0.5)
    keep = ovlp_filter(a, b)
    ref = []
    for i in range(200):
        iw = min(a[i, 2], b[i, 2]) - max(a[i, 0], b[i, 0]) + 1
        ih = min(a[i, 3], b[i, 3]) - max(a[i, 1], b[i, 1]) + 1
        if iw > 0 and ih > 0:
            ua = (a[i, 2] - a[i, 0] + 1) * (a[i, 3] - a[i, 1] + 1) + (b[i, 2] - b[i, 0] + 1) * (b[i, 3] - b[i, 1] + 1) - iw * ih
            if iw * ih / ua > 0:
                ref.append(i)
    assert 0 < len(ref) < 200 and np.array_equal(keep, np.array(ref))
    far = b + 1000.0
    assert np.array_equal(ovlp_filter(a, far), np.arange(200))       # nothing overlaps -> every pair kept (:99)
