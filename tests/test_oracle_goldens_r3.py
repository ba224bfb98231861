"""The reference's Cython overlap helpers, compiled from its own .pyx sources (oracle/build_ref.py:build_cython), pin the
oracle's restatements: tests/golden/cy_overlaps.npz holds their inputs and outputs (tools/make_goldens_r3.py).
cython_bbox.bbox_overlaps (Detectron_pytorch/lib/utils/cython_bbox.pyx:32-73) and cython_bbox_rel.bbox_pair_overlaps
(lib/utils_rel/cython_bbox_rel.pyx:37-77): float32 arithmetic, so the C restatement must match bit for bit."""
import os

import numpy as np
import pytest

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "cy_overlaps.npz"))
OV = ("rand", "dense", "touch", "big")
PO = ("rand", "dense", "touch")


@pytest.mark.parametrize("case", OV)
def test_c_oracle_bbox_overlaps_matches_the_reference_cython(case):
    from oracle import c_oracle
    got = c_oracle.bbox_overlaps(GOLD["ov_%s_a" % case], GOLD["ov_%s_b" % case])
    assert np.array_equal(got, GOLD["ov_%s_out" % case])


@pytest.mark.parametrize("case", PO)
def test_c_oracle_bbox_pair_overlaps_matches_the_reference_cython(case):
    from oracle import c_oracle
    got = c_oracle.bbox_pair_overlaps(GOLD["po_%s_a" % case], GOLD["po_%s_b" % case])
    assert np.array_equal(got, GOLD["po_%s_out" % case])


@pytest.mark.parametrize("case", OV)
def test_host_bbox_overlaps_matches_the_reference_cython(case):
    """trace_amd.evaluation.bbox_overlaps (host numpy, used by the recall / mAP code) against the same vectors."""
    from trace_amd.evaluation import bbox_overlaps
    got = bbox_overlaps(GOLD["ov_%s_a" % case], GOLD["ov_%s_b" % case])
    assert np.array_equal(got, GOLD["ov_%s_out" % case])


def test_rebuilt_reference_modules_reproduce_the_fixture():
    """Where /root/reference is present (the build container) the modules are rebuilt and must reproduce the committed
    vectors; elsewhere (the GPU box) the fixture stands alone."""
    from oracle import build_ref
    if not all(os.path.exists(p) for p in build_ref.REF_PYX.values()):
        pytest.skip("reference sources not present")
    build_ref.build_cython()
    cb, cbr = build_ref.load_cython("cython_bbox"), build_ref.load_cython("cython_bbox_rel")
    for case in OV:
        assert np.array_equal(np.asarray(cb.bbox_overlaps(GOLD["ov_%s_a" % case], GOLD["ov_%s_b" % case]), np.float32),
                              GOLD["ov_%s_out" % case])
    for case in PO:
        assert np.array_equal(np.asarray(cbr.bbox_pair_overlaps(GOLD["po_%s_a" % case], GOLD["po_%s_b" % case]), np.float32),
                              GOLD["po_%s_out" % case])
