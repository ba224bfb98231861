"""GPU parity: ROIAlign / ROIPool / NMS / box helpers through the C-ABI against
the CPU oracle (bit-exact for f32 forward / indices) and the golden fixtures."""
import numpy as np
import pytest
import torch

from conftest import synth_boxes, distinct_scores
from oracle import c_oracle, np_oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rois(rng, n, B, k=2.0, size=224.0):
    return np.concatenate([rng.integers(0, B, (n, 1)).astype(np.float32), synth_boxes(rng, n, k, size)], 1)


def test_roi_align_fwd_golden_bitexact(golden):
    from trace_amd import _C
    g = golden("roi_align_fwd.npz")
    feat = torch.from_numpy(g["feat"]).to(DEV)
    rois = torch.from_numpy(g["rois"]).to(DEV)
    for sr in (0, 2):
        for scale in (1.0 / 16, 0.25):
            out = _C.roi_align_forward(feat, rois, scale, 7, 7, sr).cpu().numpy()
            assert np.array_equal(out, g["out_sr%d_s%g" % (sr, scale)]), (sr, scale)
    out = _C.roi_align_forward(feat, rois, 1.0 / 16, 5, 3, 0).cpu().numpy()
    assert np.array_equal(out, g["out_5x3_sr0"])


@pytest.mark.parametrize("sr", [0, 2])
def test_roi_align_fwd_vs_oracle_bitexact(sr):
    from trace_amd.roi_layers import ROIAlign
    rng = np.random.default_rng(11 + sr)
    x = rng.standard_normal((3, 64, 14, 14)).astype(np.float32)
    rois = _rois(rng, 50, 3)
    out = ROIAlign((7, 7), 1 / 16., sr)(torch.from_numpy(x).to(DEV), torch.from_numpy(rois).to(DEV))
    assert np.array_equal(out.cpu().numpy(), c_oracle.roi_align_fwd(x, rois, 1 / 16., 7, 7, sr))


def test_roi_align_empty_and_bwd():
    from trace_amd.roi_layers import ROIAlign
    rng = np.random.default_rng(5)
    x = torch.from_numpy(rng.standard_normal((2, 8, 10, 12)).astype(np.float32)).to(DEV)
    out = ROIAlign((7, 7), 1 / 16., 2)(x, torch.zeros((0, 5), device=DEV))
    assert tuple(out.shape) == (0, 8, 7, 7)
    rois = _rois(rng, 9, 2, 1.5, 180.0)
    for sr in (0, 2):
        xr = x.clone().requires_grad_(True)
        y = ROIAlign((7, 7), 1 / 16., sr)(xr, torch.from_numpy(rois).to(DEV))
        g = torch.from_numpy(rng.standard_normal(tuple(y.shape)).astype(np.float32)).to(DEV)
        y.backward(g)
        ref = c_oracle.roi_align_bwd(g.cpu().numpy(), rois, 1 / 16., 7, 7, 2, 8, 10, 12, sr)
        # float atomics: order of adds differs -> tolerance, not bit equality
        np.testing.assert_allclose(xr.grad.cpu().numpy(), ref, rtol=1e-5, atol=1e-5)


def test_roi_pool_fwd_bwd():
    from trace_amd.roi_layers import ROIPool
    rng = np.random.default_rng(6)
    xn = rng.standard_normal((2, 16, 14, 14)).astype(np.float32)
    rois = _rois(rng, 12, 2)
    x = torch.from_numpy(xn).to(DEV).requires_grad_(True)
    y = ROIPool((7, 7), 1 / 16.)(x, torch.from_numpy(rois).to(DEV))
    out, am = c_oracle.roi_pool_fwd(xn, rois, 1 / 16., 7, 7)
    assert np.array_equal(y.detach().cpu().numpy(), out)
    g = rng.standard_normal(out.shape).astype(np.float32)
    y.backward(torch.from_numpy(g).to(DEV))
    ref = c_oracle.roi_pool_bwd(g, am, rois, 7, 7, 2, 16, 14, 14)
    np.testing.assert_allclose(x.grad.cpu().numpy(), ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("sr", [0, 2])
@pytest.mark.parametrize("shape", [(2, 256, 14, 14), (2, 128, 20, 30), (2, 72, 14, 14)])
def test_roi_align_nhwc_bf16(sr, shape):
    """Channels-last bf16 variant: same sampling rule, bf16 I/O -> rtol 2e-2 (north_star).  C % 128 == 0 takes the
    separable one-pass kernel, other channel counts the per-bin kernel; degenerate, out-of-map and whole-map rois."""
    from trace_amd import ops
    rng = np.random.default_rng(21)
    x = rng.standard_normal(shape).astype(np.float32)
    xb = torch.from_numpy(x).to(DEV).to(torch.bfloat16)
    xr = xb.float().cpu().numpy()  # oracle sees the bf16-rounded input
    rois = _rois(rng, 33, 2)
    extra = np.array([[0, 10, 10, 10, 10], [1, -40, -40, 30, 30], [1, 200, 200, 300, 300], [0, 0, 0, 223, 223],
                      [1, 500, 500, 600, 600], [0, -100, 50, -50, 90], [0, 0, 0, 479, 319], [1, 3, 100, 400, 110]], np.float32)
    rois = np.concatenate([rois, extra], 0)
    out = ops.roi_align_nhwc(xb.permute(0, 2, 3, 1).contiguous(), torch.from_numpy(rois).to(DEV), 1 / 16., 7, sr)
    ref = c_oracle.roi_align_fwd(xr, rois, 1 / 16., 7, 7, sr)
    got = out.float().permute(0, 3, 1, 2).cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("sr", [0, 2])
def test_roi_align_avg_nhwc_bf16(sr):
    """ROIAlign fused with the 7x7 mean (box_head3d.py:59-61): the separable pixel-weight kernel against
    the pinned ROIAlign oracle followed by a mean; includes degenerate, out-of-map and whole-map rois."""
    from trace_amd import ops
    rng = np.random.default_rng(22)
    x = rng.standard_normal((3, 64, 14, 14)).astype(np.float32)
    xb = torch.from_numpy(x).to(DEV).to(torch.bfloat16)
    xr = xb.float().cpu().numpy()
    rois = _rois(rng, 40, 3)
    extra = np.array([[0, 10, 10, 10, 10], [1, -40, -40, 30, 30], [2, 200, 200, 300, 300], [0, 0, 0, 223, 223],
                      [1, 500, 500, 600, 600], [2, -100, 50, -50, 90]], np.float32)
    rois = np.concatenate([rois, extra], 0)
    out = ops.roi_align_avg_nhwc(xb.permute(0, 2, 3, 1).contiguous(), torch.from_numpy(rois).to(DEV), 1 / 16., 7, sr)
    ref = c_oracle.roi_align_fwd(xr, rois, 1 / 16., 7, 7, sr).mean(axis=(2, 3))
    np.testing.assert_allclose(out.float().cpu().numpy(), ref, rtol=2e-2, atol=1e-2)


def test_roi_ops_float64_dispatch(golden):
    """The reference's extension dispatches its ROI ops on float and double (ROIAlign_cuda.cu:283,330, ROIPool_cuda.cu:
    129,178): `_C.roi_align_forward` in float64 is bit-exact against the golden of the reference's own compiled CPU code
    and against the oracle on fresh inputs; backward and ROIPool against the double oracle (sums of atomics: 1e-12);
    mixed dtypes and half precision are refused as a RuntimeError."""
    from trace_amd import _C
    g = golden("roi_align_fwd_f64.npz")
    feat, rois = torch.from_numpy(g["feat"]).to(DEV), torch.from_numpy(g["rois"]).to(DEV)
    for sr in (0, 2):
        for scale in (1.0 / 16, 0.25):
            out = _C.roi_align_forward(feat, rois, scale, 7, 7, sr)
            assert out.dtype == torch.float64 and np.array_equal(out.cpu().numpy(), g["out_sr%d_s%g" % (sr, scale)]), (sr, scale)
    assert np.array_equal(_C.roi_align_forward(feat, rois, 1.0 / 16, 5, 3, 0).cpu().numpy(), g["out_5x3_sr0"])
    rng = np.random.default_rng(64)
    x = rng.standard_normal((3, 5, 20, 17))
    r = _rois(rng, 30, 3).astype(np.float64) + rng.uniform(0, 1e-6, (30, 5)) * np.array([0, 1, 1, 1, 1.0])
    xd, rd = torch.from_numpy(x).to(DEV), torch.from_numpy(r).to(DEV)
    for sr in (0, 3):
        assert np.array_equal(_C.roi_align_forward(xd, rd, 1 / 16., 7, 7, sr).cpu().numpy(), c_oracle.roi_align_fwd(x, r, 1 / 16., 7, 7, sr))
        gy = rng.standard_normal((30, 5, 7, 7))
        gin = _C.roi_align_backward(torch.from_numpy(gy).to(DEV), rd, 1 / 16., 7, 7, 3, 5, 20, 17, sr)
        assert gin.dtype == torch.float64
        np.testing.assert_allclose(gin.cpu().numpy(), c_oracle.roi_align_bwd(gy, r, 1 / 16., 7, 7, 3, 5, 20, 17, sr), rtol=1e-12, atol=1e-12)
    po, am = _C.roi_pool_forward(xd, rd, 1 / 16., 7, 7)
    wo, wa = c_oracle.roi_pool_fwd(x, r, 1 / 16., 7, 7)
    assert po.dtype == torch.float64 and np.array_equal(po.cpu().numpy(), wo) and np.array_equal(am.cpu().numpy(), wa)
    gy = rng.standard_normal((30, 5, 7, 7))
    gp = _C.roi_pool_backward(torch.from_numpy(gy).to(DEV), xd, rd, am, 1 / 16., 7, 7, 3, 5, 20, 17)
    np.testing.assert_allclose(gp.cpu().numpy(), c_oracle.roi_pool_bwd(gy, wa, r, 7, 7, 3, 5, 20, 17), rtol=1e-12, atol=1e-12)
    with pytest.raises(RuntimeError):
        _C.roi_align_forward(xd, rd.float(), 1 / 16., 7, 7, 0)
    with pytest.raises(RuntimeError):
        _C.roi_align_forward(xd.half(), rd.half(), 1 / 16., 7, 7, 0)
    # the layer modules pass the dtype through (roi_layers/roi_align.py:11-59): gradient check in double
    from trace_amd.roi_layers import ROIAlign
    xs = torch.from_numpy(rng.standard_normal((1, 2, 8, 8))).to(DEV).requires_grad_(True)
    rr = torch.tensor([[0, 8.0, 12.0, 90.0, 100.0], [0, 20.0, 4.0, 60.0, 70.0]], dtype=torch.float64, device=DEV)
    assert torch.autograd.gradcheck(lambda t: ROIAlign((3, 3), 1 / 16., 2)(t, rr), (xs,), eps=1e-6, atol=1e-8)


def test_roi_align_full_size_constant_property():
    """BASELINE cfg5-sized call (2100 rois x 2048 ch): a constant map pools to the constant."""
    from trace_amd import _C
    x = torch.full((8, 2048, 14, 14), 1.5, device=DEV)
    rng = np.random.default_rng(2)
    rois = torch.from_numpy(_rois(rng, 2100, 8)).to(DEV)
    out = _C.roi_align_forward(x, rois, 1 / 16., 7, 7, 0)
    assert out.shape == (2100, 2048, 7, 7)
    # inside-the-map boxes: every sample is a convex combination of 1.5 -> within rounding of 1.5
    assert float((out - 1.5).abs().max()) < 1e-5


def test_nms_golden_bitexact(golden):
    """Golden keeps come from the reference's own CPU kernel (nms_cpu.cpp, `>=`): NMS_GE.  The default mode of the
    device binding is the CUDA rule (`>`), see test_nms_ties_and_exact_threshold."""
    from trace_amd import _C
    from trace_amd.roi_layers import nms
    g = golden("nms.npz")
    names = sorted({k.rsplit("_", 1)[0] for k in g.files})
    for n in names:
        keep = nms(torch.from_numpy(g[n + "_boxes"]).to(DEV), torch.from_numpy(g[n + "_scores"]).to(DEV),
                   float(g[n + "_thresh"]), _C.NMS_GE)
        assert keep.dtype == torch.int64
        assert np.array_equal(keep.cpu().numpy(), g[n + "_keep"]), n


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 129, 300, 1000, 6000])
@pytest.mark.parametrize("mode", [0, 1])
def test_nms_vs_oracle(n, mode):
    from trace_amd import _C
    rng = np.random.default_rng(100 + n)
    boxes = synth_boxes(rng, n, 2.0 if n < 2000 else 1.0)
    scores = distinct_scores(rng, n)
    thr = 0.7 if n >= 1000 else 0.5
    keep = _C.nms(torch.from_numpy(boxes).to(DEV), torch.from_numpy(scores).to(DEV), thr, mode)
    assert np.array_equal(keep.cpu().numpy(), c_oracle.nms(boxes, scores, thr, mode))


def test_nms_ties_and_exact_threshold(golden):
    from trace_amd import _C
    g = golden("nms.npz")
    b = torch.from_numpy(g["eq_boxes"]).to(DEV)
    s = torch.from_numpy(g["eq_scores"]).to(DEV)
    assert _C.nms(b, s, 0.5, _C.NMS_GE).tolist() == [0, 2]
    assert _C.nms(b, s, 0.5, _C.NMS_GT).tolist() == [0, 1, 2]
    # score ties: documented policy = lower index first (same as the oracle)
    rng = np.random.default_rng(9)
    boxes = synth_boxes(rng, 200)
    scores = (rng.integers(0, 5, 200) / 5.0).astype(np.float32)
    keep = _C.nms(torch.from_numpy(boxes).to(DEV), torch.from_numpy(scores).to(DEV), 0.5)     # default = the CUDA rule (>)
    assert np.array_equal(keep.cpu().numpy(), c_oracle.nms(boxes, scores, 0.5, 1))


def test_nms_idempotent_full_size():
    """Size-independent property at the RPN size: NMS of the kept set keeps everything."""
    from trace_amd import _C
    rng = np.random.default_rng(77)
    boxes = torch.from_numpy(synth_boxes(rng, 6000, 1.0)).to(DEV)
    scores = torch.from_numpy(distinct_scores(rng, 6000)).to(DEV)
    keep = _C.nms(boxes, scores, 0.7)
    assert 0 < keep.numel() < 6000
    assert bool((keep[1:] > keep[:-1]).all())
    again = _C.nms(boxes[keep].contiguous(), scores[keep].contiguous(), 0.7)
    assert again.numel() == keep.numel()


def test_box_helpers_bitexact():
    from trace_amd import boxes as tb
    rng = np.random.default_rng(31)
    b = synth_boxes(rng, 5000, 8.0, 1400.0)
    b[:50, 2:] = b[:50, :2] - rng.uniform(0, 30, (50, 2)).astype(np.float32)  # degenerate / negative extents
    exact = np.array([[0, 0, 223, 223], [0, 0, 111, 111], [0, 0, 447, 447], [0, 0, 55, 55], [3, 3, 2, 2]], np.float32)
    b = np.concatenate([b, exact], 0)
    lv = tb.map_rois_to_fpn_levels(torch.from_numpy(b).to(DEV), 2, 5).cpu().numpy()
    assert np.array_equal(lv, np_oracle.map_rois_to_fpn_levels(b.copy(), 2, 5))
    rois = np.concatenate([np.zeros((len(b), 1), np.float32), b], 1)
    blobs = tb.add_multilevel_roi_blobs({}, "rois", torch.from_numpy(rois).to(DEV), torch.from_numpy(lv).to(DEV), 2, 5)
    ref = np_oracle.add_multilevel_roi_blobs({}, "rois", rois, lv, 2, 5)
    for k in ref:
        assert np.array_equal(blobs[k].cpu().numpy(), ref[k]), k
    q = synth_boxes(rng, 37)
    a = synth_boxes(rng, 211)
    assert np.array_equal(tb.bbox_overlaps(torch.from_numpy(a).to(DEV), torch.from_numpy(q).to(DEV)).cpu().numpy(),
                          c_oracle.bbox_overlaps(a, q))
    a2 = synth_boxes(rng, 211)
    assert np.array_equal(tb.bbox_pair_overlaps(torch.from_numpy(a).to(DEV), torch.from_numpy(a2).to(DEV)).cpu().numpy(),
                          c_oracle.bbox_pair_overlaps(a, a2))
    for n in (1, 2, 20, 100):
        r = _rois(rng, n, 1)
        r[:, 0] = 0
        s, o, rel = tb.relpn_pairs(torch.from_numpy(r).to(DEV))
        rs, ro, rrel = np_oracle.relpn_pairs(r)
        assert np.array_equal(s.cpu().numpy(), rs) and np.array_equal(o.cpu().numpy(), ro)
        assert np.array_equal(rel.cpu().numpy(), rrel.astype(np.float32))


@pytest.mark.parametrize("case", ["rand", "dense", "touch", "big"])
def test_bbox_overlaps_vs_reference_cython_golden(case):
    """trace_bbox_overlaps against the outputs of the reference's own cython_bbox.bbox_overlaps (compiled from its .pyx,
    tests/golden/cy_overlaps.npz): bit for bit."""
    import os
    from trace_amd import boxes as tb
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "cy_overlaps.npz"))
    a, b = G["ov_%s_a" % case], G["ov_%s_b" % case]
    got = tb.bbox_overlaps(torch.from_numpy(a).to(DEV), torch.from_numpy(b).to(DEV)).cpu().numpy()
    assert np.array_equal(got, G["ov_%s_out" % case])


@pytest.mark.parametrize("case", ["rand", "dense", "touch"])
def test_bbox_pair_overlaps_vs_reference_cython_golden(case):
    """trace_bbox_pair_overlaps against cython_bbox_rel.bbox_pair_overlaps' own outputs: bit for bit."""
    import os
    from trace_amd import boxes as tb
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "cy_overlaps.npz"))
    a, b = G["po_%s_a" % case], G["po_%s_b" % case]
    got = tb.bbox_pair_overlaps(torch.from_numpy(a).to(DEV), torch.from_numpy(b).to(DEV)).cpu().numpy()
    assert np.array_equal(got, G["po_%s_out" % case])
