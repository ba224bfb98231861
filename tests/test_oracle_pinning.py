"""The CPU oracle (oracle/trace_oracle.c) against golden vectors produced by the
reference's own compiled CPU sources (tools/make_goldens.py -> oracle/_ref)."""
import numpy as np
import pytest

from oracle import c_oracle, np_oracle
from conftest import synth_boxes, distinct_scores


def test_roi_align_fwd_f64_matches_reference_golden(golden):
    """The double half of the reference's dtype dispatch (cpu/ROIAlign_cpu.cpp:243): the oracle built over double
    reproduces the outputs of the reference's own compiled code bit for bit (tools/make_goldens_r3.py)."""
    g = golden("roi_align_fwd_f64.npz")
    feat, rois = g["feat"], g["rois"]
    assert feat.dtype == np.float64 and rois.dtype == np.float64
    for sr in (0, 2):
        for scale in (1.0 / 16, 0.25):
            out = c_oracle.roi_align_fwd(feat, rois, scale, 7, 7, sr)
            assert out.dtype == np.float64 and np.array_equal(out, g["out_sr%d_s%g" % (sr, scale)]), (sr, scale)
    assert np.array_equal(c_oracle.roi_align_fwd(feat, rois, 1.0 / 16, 5, 3, 0), g["out_5x3_sr0"])
    # a float32 computation does not reproduce them: the fixture does test the double path
    lo = c_oracle.roi_align_fwd(feat.astype(np.float32), rois.astype(np.float32), 1.0 / 16, 7, 7, 0)
    assert not np.array_equal(lo.astype(np.float64), g["out_sr0_s0.0625"])


def test_roi_align_fwd_matches_reference_golden(golden):
    g = golden("roi_align_fwd.npz")
    feat, rois = g["feat"], g["rois"]
    for sr in (0, 2):
        for scale in (1.0 / 16, 0.25):
            out = c_oracle.roi_align_fwd(feat, rois, scale, 7, 7, sr)
            assert np.array_equal(out, g["out_sr%d_s%g" % (sr, scale)]), (sr, scale)
    assert np.array_equal(c_oracle.roi_align_fwd(feat, rois, 1.0 / 16, 5, 3, 0), g["out_5x3_sr0"])


def test_nms_matches_reference_golden(golden):
    g = golden("nms.npz")
    names = sorted({k.rsplit("_", 1)[0] for k in g.files})
    assert len(names) >= 6
    for n in names:
        keep = c_oracle.nms(g[n + "_boxes"], g[n + "_scores"], float(g[n + "_thresh"]), 0)
        assert np.array_equal(keep, g[n + "_keep"]), n


def test_nms_ge_vs_gt_at_exact_threshold(golden):
    g = golden("nms.npz")
    b, s = g["eq_boxes"], g["eq_scores"]
    assert list(c_oracle.nms(b, s, 0.5, 0)) == [0, 2]       # >= : IoU==0.5 suppressed (cython_nms.pyx:84)
    assert list(c_oracle.nms(b, s, 0.5, 1)) == [0, 1, 2]    # >  : kept (cuda/nms.cu:60)


def test_live_reference_if_present():
    """When oracle/_ref was built (build container / shipped .so), compare live on fresh seeds."""
    from oracle.build_ref import load_ref
    ref = load_ref()
    if ref is None:
        pytest.skip("oracle/_ref not built")
    import torch
    rng = np.random.default_rng(7)
    feat = rng.standard_normal((3, 8, 20, 17)).astype(np.float32)
    boxes = synth_boxes(rng, 40, 3.0, 300.0)
    rois = np.concatenate([rng.integers(0, 3, (40, 1)).astype(np.float32), boxes], 1)
    for sr in (0, 1, 2, 3):
        a = c_oracle.roi_align_fwd(feat, rois, 1 / 16., 7, 7, sr)
        b = ref.roi_align_forward(torch.from_numpy(feat), torch.from_numpy(rois), 1 / 16., 7, 7, sr).numpy()
        assert np.array_equal(a, b)
    for n in (1, 2, 65, 500):
        bx = synth_boxes(rng, n)
        sc = distinct_scores(rng, n)
        a = c_oracle.nms(bx, sc, 0.5, 0)
        b = ref.nms(torch.from_numpy(bx), torch.from_numpy(sc), 0.5).numpy()
        assert np.array_equal(a, b)


def test_roi_align_bwd_is_adjoint_of_fwd():
    """No CPU backward exists in the reference (ROIAlign.h:44): pin the oracle's
    backward as the exact adjoint of the (pinned) forward: <fwd(x), g> == <x, bwd(g)>."""
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 3, 9, 11)).astype(np.float32)
    rois = np.concatenate([rng.integers(0, 2, (6, 1)).astype(np.float32), synth_boxes(rng, 6, 1.5, 160.0)], 1)
    for sr in (0, 2):
        y = c_oracle.roi_align_fwd(x, rois, 1 / 16., 7, 7, sr)
        g = rng.standard_normal(y.shape).astype(np.float32)
        gx = c_oracle.roi_align_bwd(g, rois, 1 / 16., 7, 7, 2, 3, 9, 11, sr)
        lhs = float((y.astype(np.float64) * g).sum())
        rhs = float((x.astype(np.float64) * gx).sum())
        assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(lhs))


def test_roi_pool_fwd_bwd_consistency():
    rng = np.random.default_rng(4)
    x = rng.standard_normal((2, 3, 12, 12)).astype(np.float32)
    rois = np.concatenate([rng.integers(0, 2, (5, 1)).astype(np.float32), synth_boxes(rng, 5, 1.5, 190.0)], 1)
    out, am = c_oracle.roi_pool_fwd(x, rois, 1 / 16., 7, 7)
    flat = x.reshape(2, 3, -1)
    for n in range(5):
        b = int(rois[n, 0])
        for c in range(3):
            idx = am[n, c].ravel()
            vals = out[n, c].ravel()
            for i, v in zip(idx, vals):
                if i >= 0:
                    assert flat[b, c, i] == v
                else:
                    assert v == 0
    g = np.ones_like(out)
    gx = c_oracle.roi_pool_bwd(g, am, rois, 7, 7, 2, 3, 12, 12)
    assert gx.sum() == (am >= 0).sum()


def test_fpn_levels_known_answers():
    # fpn.py:23: lvl = floor(4 + log2(sqrt(area)/224 + 1e-6)) clipped to [2,5]
    boxes = np.array([[0, 0, 223, 223],    # 224x224 -> 4
                      [0, 0, 111, 111],    # 112x112 -> 3
                      [0, 0, 447, 447],    # 448 -> 5
                      [0, 0, 2000, 2000],  # clip 5
                      [0, 0, 5, 5],        # clip 2
                      [10, 10, 5, 5]],     # negative w,h: area positive product (-4*-4=16) -> clip 2
                     np.float32)
    lv = np_oracle.map_rois_to_fpn_levels(boxes, 2, 5)
    assert lv.dtype == np.float32
    assert list(lv) == [4, 3, 5, 5, 2, 2]
    blobs = {}
    rois = np.concatenate([np.zeros((6, 1), np.float32), boxes], 1)
    np_oracle.add_multilevel_roi_blobs(blobs, "rois", rois, lv, 2, 5)
    assert blobs["rois_idx_restore_int32"].dtype == np.int32
    stacked = np.vstack([blobs["rois_fpn%d" % l] for l in range(2, 6)])
    assert np.array_equal(stacked[blobs["rois_idx_restore_int32"]], rois)


def test_relpn_pairs_and_tree_invariants():
    rng = np.random.default_rng(5)
    for n in (1, 2, 3, 10, 21):
        boxes = synth_boxes(rng, n)
        rois = np.concatenate([np.zeros((n, 1), np.float32), boxes], 1)
        s, o, rel = np_oracle.relpn_pairs(rois)
        P = n * (n - 1) if n > 1 else 1
        assert len(s) == len(o) == P == rel.shape[0]
        if n > 1:
            assert (s != o).all()
        layers, tree = np_oracle.node_cluster(boxes)
        n_tree = tree.shape[0]
        assert np.array_equal(tree[:n], boxes)
        # every node except the root has exactly one parent; parents are hulls
        parent = -np.ones(n_tree, np.int64)
        for layer in layers:
            for fa, sons in layer.items():
                sons = np.array(sons)
                assert (parent[sons] == -1).all()
                parent[sons] = fa
                assert np.array_equal(tree[fa, :2], tree[sons, :2].min(0))
                assert np.array_equal(tree[fa, 2:], tree[sons, 2:].max(0))
        assert (parent[:-1] >= 0).all() and parent[-1] == -1
        A, A_T, link = np_oracle.get_A_and_link(n_tree, layers)
        assert np.array_equal(A.T, A_T)
        if n > 1:
            lca = np_oracle.get_lca(s, o, link)
            assert (lca >= n).all() and (lca < n_tree).all()
