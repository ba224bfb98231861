"""GPU parity of the detected-box branch (model_builder_rel.py:1212-1340: box decode + clip, per-class
NMS in one grouped pass, detections-per-image cap) through the C-ABI, against the CPU oracle and the
golden vectors produced by the reference's own Python (tests/golden/py_det_branch.npz, py_box_glue.npz)."""
import json

import numpy as np
import pytest
import torch

from conftest import synth_boxes, distinct_scores
from oracle import c_oracle, np_oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def jload(a):
    return json.loads(bytes(a.tobytes()).decode())


@pytest.mark.parametrize("N,groups", [(1, 1), (64, 3), (777, 8), (3000, 35), (9000, 5)])
def test_nms_grouped_equals_per_group_nms(N, groups):
    """Keeps of one grouped pass == union of the oracle's NMS on every group alone (bit-exact indices).
    N=9000 also covers the > 8192-box sort (128 KiB of LDS)."""
    from trace_amd import boxes as B
    rng = np.random.default_rng(N)
    bx = synth_boxes(rng, N, 2.0)
    sc = distinct_scores(rng, N)
    gr = rng.integers(0, groups, N).astype(np.int32)
    keep = B.nms_grouped(torch.from_numpy(bx).to(DEV), torch.from_numpy(sc).to(DEV), torch.from_numpy(gr).to(DEV), 0.5)
    ref = []
    for g in range(groups):
        idx = np.where(gr == g)[0]
        if idx.size:
            ref.append(idx[c_oracle.nms(np.ascontiguousarray(bx[idx]), np.ascontiguousarray(sc[idx]), 0.5, 0)])
    ref = np.sort(np.concatenate(ref))
    assert np.array_equal(keep.cpu().numpy(), ref)


def test_nms_grouped_ragged_groups_and_sparse_ids():
    """One class with 2500 boxes (a 40-block chain), 60 single-box classes, classes of 63 / 64 / 65 boxes that start and end
    on and off 64-box block boundaries, sparse class ids up to 2^18 - 2: same keeps as per-class NMS."""
    from trace_amd import boxes as B
    rng = np.random.default_rng(77)
    sizes = [2500, 63, 64, 65, 1, 1, 129] + [1] * 58 + [300]
    ids = rng.choice(np.arange(1, 2 ** 18 - 1), size=len(sizes), replace=False).astype(np.int32)
    ids[0], ids[-1] = 2 ** 18 - 2, 0
    gr = np.concatenate([np.full(n, i, np.int32) for n, i in zip(sizes, ids)])
    perm = rng.permutation(gr.shape[0])
    gr = gr[perm]
    N = gr.shape[0]
    bx = synth_boxes(rng, N, 1.5)
    sc = distinct_scores(rng, N)
    keep = B.nms_grouped(torch.from_numpy(bx).to(DEV), torch.from_numpy(sc).to(DEV), torch.from_numpy(gr).to(DEV), 0.5)
    ref = []
    for g in ids:
        idx = np.where(gr == g)[0]
        ref.append(idx[c_oracle.nms(np.ascontiguousarray(bx[idx]), np.ascontiguousarray(sc[idx]), 0.5, 0)])
    assert np.array_equal(keep.cpu().numpy(), np.sort(np.concatenate(ref)))


def test_nms_large_single_group_vs_oracle():
    from trace_amd import _C
    rng = np.random.default_rng(3)
    N = 12000
    bx = synth_boxes(rng, N, 1.0)
    sc = distinct_scores(rng, N)
    keep = _C.nms(torch.from_numpy(bx).to(DEV), torch.from_numpy(sc).to(DEV), 0.7, _C.NMS_GE)
    assert np.array_equal(keep.cpu().numpy(), c_oracle.nms(bx, sc, 0.7, 0))


def test_bbox_transform_clip_vs_reference_golden(golden):
    from trace_amd import boxes as B
    g = golden("py_box_glue.npz")
    out = B.bbox_transform_clip(torch.from_numpy(g["bt_boxes"]).to(DEV), torch.from_numpy(g["bt_deltas"]).to(DEV),
                                tuple(g["bt_weights"].tolist()), tuple(g["bt_im_shape"].tolist()),
                                float(g["bt_xform_clip"]))
    # f32 throughout as the reference; only expf may differ from numpy's by an ulp -> 1e-5 of the image size
    np.testing.assert_allclose(out.cpu().numpy(), g["bt_clipped"], rtol=0, atol=600 * 1e-5)


def test_prepare_det_rois_vs_reference_golden(golden):
    from trace_amd import boxes as B
    g = golden("py_det_branch.npz")
    c = jload(g["cfg"])
    for key in jload(g["cases"]):
        dr, dl, ds, di = B.prepare_det_rois(torch.from_numpy(g[key + "_rois"]).to(DEV),
                                            torch.from_numpy(g[key + "_cls_scores"]).to(DEV),
                                            torch.from_numpy(g[key + "_bbox_pred"]).to(DEV), g[key + "_im_info"],
                                            score_thresh=c["score_thresh"], nms_thresh=c["nms"],
                                            detections_per_im=c["detections_per_im"], batch_size=1,
                                            reg_weights=tuple(c["reg_weights"]))
        assert np.array_equal(di.cpu().numpy(), g[key + "_out_inds"]), key        # which rois survive: bit-exact
        assert np.array_equal(dl.cpu().numpy(), g[key + "_out_det_labels"]), key
        assert np.array_equal(ds.cpu().numpy(), g[key + "_out_det_scores"]), key
        np.testing.assert_allclose(dr.cpu().numpy(), g[key + "_out_det_rois"], rtol=0, atol=800 * 1e-5)


def test_box_results_vs_oracle_random():
    from trace_amd import boxes as B
    rng = np.random.default_rng(9)
    R, C = 500, 36
    scores = rng.random((R, C)).astype(np.float32) ** 6          # most below the 0.05 threshold
    bx = np.concatenate([synth_boxes(rng, R, 3.0, 800.0) for _ in range(C)], 1)
    s, b, l, ii = B.box_results_with_nms_and_limit(torch.from_numpy(scores).to(DEV), torch.from_numpy(bx).to(DEV),
                                                   0.05, 0.5, 100)
    rs, rb, rl, ri = np_oracle.box_results_with_nms_and_limit(scores, bx, C, 0.05, 0.5, 100)
    assert np.array_equal(ii.cpu().numpy(), ri) and np.array_equal(l.cpu().numpy(), rl)
    assert np.array_equal(s.cpu().numpy(), rs) and np.array_equal(b.cpu().numpy(), rb)
