"""BASELINE.json's configurations at their OWN sizes (the smaller parity cases live in test_head_gpu.py / test_e2e_gpu.py):
  configs[1]  VidVRD setting: 8 clips x 32 frames x 224^2, 20 objects / 380 pairs per clip (the bench workload)
  configs[2]  Action Genome setting: ~50 objects, ~2.5k pairs, USE_OVLP_FILTER, persons-only subjects + the NMS stress
              SURVEY.md 8(d) names (6 000 RPN boxes @0.7, 36 classes x <=300 boxes @0.5)
  configs[4]  long clip: 64 frames, 100 objects, 9 900 pairs (segmented device ranking)
The relation head is checked teacher-forced against the fp32 oracle (oracle head on the HIP path's own temporal maps:
the oracle backbone at these sizes would take minutes per clip; clip -> triplets without teacher forcing is
tests/test_e2e_gpu.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel_l2(got, ref):
    got, ref = got.float().cpu(), ref.float().cpu()
    return float((got - ref).norm() / (ref.norm() + 1e-12))


def build(cfg=None, seed=1):
    from trace_amd.model_builder_rel import Generalized_RCNN
    from trace_amd.synthetic import randomize_model
    from oracle.i3d_ref import calibrate_bn
    torch.manual_seed(seed)
    m = Generalized_RCNN(cfg)
    randomize_model(m, seed)
    calib = torch.randn((1, 3, 16, 112, 112), generator=torch.Generator().manual_seed(1)) * 50
    m.Prd_RCNN.load_state_dict(calibrate_bn(m.Prd_RCNN.state_dict(), calib))
    return m.eval()


def check_clip(m, sd, outs, b, blob, feat, rois, prd_k, poff, labels=None, scores=None, **okw):
    from oracle.head_ref import head_forward
    from oracle.np_oracle import topk_triplets
    from tests.test_head_gpu import check_topk
    tm = m._debug["tmaps"][b].float().permute(0, 3, 1, 2).cpu()
    ref = head_forward(sd, blob[b:b + 1], feat[b:b + 1], tm, rois[b],
                       det_labels_in=None if labels is None else labels[b],
                       det_scores_in=None if scores is None else scores[b], **okw)
    o = outs[b]
    P = len(ref["sbj_inds"])
    assert np.array_equal(o["sbj_inds"], ref["sbj_inds"]) and np.array_equal(o["obj_inds"], ref["obj_inds"])
    e = rel_l2(m._debug["logits"][poff:poff + P], ref["logits"])
    assert e < 2e-2, (b, e)
    # relation scores: 2e-2 of the score scale over the clip, and no single score of the (up to 1.3 M) off by more than 4e-2
    assert rel_l2(o["prd_scores"], ref["prd_scores"]) < 2e-2
    np.testing.assert_allclose(o["prd_scores"].float().cpu().numpy(), ref["prd_scores"].numpy(), atol=4e-2)
    rs = ref["det_scores"]
    sc, pair, prd = topk_triplets(ref["prd_scores"].numpy(), rs[ref["sbj_inds"]], rs[ref["obj_inds"]], prd_k, 100)
    want = set(zip(pair.tolist(), prd.tolist()))
    got = set(zip(o["top_pair"].cpu().tolist(), o["top_prd"].cpu().tolist()))
    check_topk(want, got, tag="clip %d P=%d prd_k=%d" % (b, P, prd_k))
    return P


def test_config1_vidvrd_batch8_32f_20_objects():
    """configs[1] at B=8, T=32, n=20, through the whole-step HIP graph as the bench runs it; oracle head on 2 of the 8
    clips (teacher-forced), the other clips checked for equality with a B=1 run of the same clip."""
    from trace_amd.synthetic import make_inputs
    from tests.test_e2e_gpu import synth_labels_scores
    m = build()
    B, T, n = 8, 32, 20
    frames, blob, feat, rois = make_inputs(B, T, n, 224, seed=1)
    labels, scores = synth_labels_scores(rois, m.cfg.NUM_CLASSES, 11)
    sd = {k: v.detach().float().cpu() for k, v in m.state_dict().items()}
    m.to(DEV)
    outs = m.forward_clips(frames.to(DEV), blob.to(DEV), feat.to(DEV), rois, topk=100, prd_k=2, det_labels=labels,
                           det_scores=scores)
    assert len(outs) == B and m._debug["tmaps"].shape[:2] == (B, 4)
    for b in (0, 5):
        P = check_clip(m, sd, outs, b, blob, feat, rois, 2, b * n * (n - 1), labels, scores)
        assert P == n * (n - 1) == 380
    eager = [{k: o[k].clone() for k in ("prd_scores", "top_pair", "top_prd", "top_scores")} for o in outs]
    m.graph_step = True
    for _ in range(2):
        g = m.forward_clips(frames.to(DEV), blob.to(DEV), feat.to(DEV), rois, topk=100, prd_k=2, det_labels=labels,
                            det_scores=scores)
        for o, w in zip(g, eager):
            for k in w:
                assert torch.equal(o[k], w[k]), k


def test_uint8_frames_give_the_f32_clips_triplets():
    """The harness hands over uint8-origin frames (get_frames_blob, test_engine_rel.py:420-520): decoded frames
    (B,T,h,w,3) with Prd_RCNN.pixel_means == the f32 clip of the same mean-subtracted pixels, bit for bit, eager and
    through the whole-step graph."""
    from trace_amd.synthetic import make_inputs
    m = build()
    B, T, n = 2, 8, 10
    _, blob, feat, rois = make_inputs(B, T, n, 224, seed=4)
    g = torch.Generator().manual_seed(4)
    frames_u8 = torch.randint(0, 256, (B, T, 224, 224, 3), generator=g, dtype=torch.uint8)
    means = [102.9801, 115.9465, 122.7717]
    clip = (frames_u8.float() - torch.tensor(means)).permute(0, 1, 4, 2, 3).contiguous()
    m.to(DEV)
    m.Prd_RCNN.pixel_means = means
    ref = m.forward_clips(clip.to(DEV), blob.to(DEV), feat.to(DEV), rois, topk=100, prd_k=2)
    ref = [{k: o[k].clone() for k in ("prd_scores", "top_pair", "top_prd", "top_scores")} for o in ref]
    for graph in (False, True):
        m.graph_step = graph
        got = m.forward_clips(frames_u8.to(DEV), blob.to(DEV), feat.to(DEV), rois, topk=100, prd_k=2)
        for o, w in zip(got, ref):
            for k in w:
                assert torch.equal(o[k], w[k]), (graph, k)


def test_config2_action_genome_50_objects_overlap_filter():
    """configs[2]: AG dims (37 / 26 classes, D = 1024), n = 50 objects, USE_OVLP_FILTER, persons-only subjects."""
    from trace_amd.config import ag_config
    from trace_amd.synthetic import make_inputs
    m = build(ag_config(), seed=2)
    B, T, n = 1, 32, 50
    frames, blob, feat, rois = make_inputs(B, T, n, 224, seed=31)
    rng = np.random.default_rng(8)
    labels = [rng.integers(2, 37, n).astype(np.int32)]
    labels[0][[0, 7, 19, 33]] = 1                                      # four persons
    scores = [(rng.permutation(n) / n * 0.9 + 0.05).astype(np.float32)]
    sd = {k: v.detach().float().cpu() for k, v in m.state_dict().items()}
    m.to(DEV)
    outs = m.forward_clips(frames.to(DEV), blob.to(DEV), feat.to(DEV), rois, topk=100, prd_k=6, det_labels=labels,
                           det_scores=scores)
    P = check_clip(m, sd, outs, 0, blob, feat, rois, 6, 0, labels, scores, dataset_name="ag", use_ovlp_filter=True,
                   groups=4)
    assert 0 < P <= 4 * 49 and outs[0]["prd_scores"].shape[1] == 26
    assert set(np.unique(outs[0]["sbj_inds"]).tolist()) <= {0, 7, 19, 33}
    # all-pairs variant of the same setting (~2.5k pairs): no persons among the labels -> every ordered pair (relpn_heads.py:79-84)
    labels2 = [np.where(labels[0] == 1, 2, labels[0]).astype(np.int32)]
    outs2 = m.forward_clips(frames.to(DEV), blob.to(DEV), feat.to(DEV), rois, topk=100, prd_k=6, det_labels=labels2,
                            det_scores=scores)
    P2 = check_clip(m, sd, outs2, 0, blob, feat, rois, 6, 0, labels2, scores, dataset_name="ag", use_ovlp_filter=True,
                    groups=4)
    assert 400 < P2 <= 2450


def test_config2_nms_stress_rpn_and_per_class():
    """6 000 RPN boxes @0.7 (generate_proposals.py:164) and 36 classes x <= 300 boxes @0.5 in one grouped
    pass (model_builder_rel.py:1286-1313); bit-exact keep lists (Cython / CPU rule `>=`)."""
    from oracle import c_oracle
    from trace_amd import _C
    from trace_amd import boxes as tb
    from trace_amd.synthetic import synth_boxes
    rng = np.random.default_rng(12)
    n = 6000
    bx = synth_boxes(rng, n, 3.0, 800.0)
    sc = (rng.permutation(n) / n).astype(np.float32)
    keep = _C.nms(torch.from_numpy(bx).to(DEV), torch.from_numpy(sc).to(DEV), 0.7, _C.NMS_GE)
    want = c_oracle.nms(bx, sc, 0.7, 0)
    assert np.array_equal(keep.cpu().numpy(), want) and len(want) > 100
    per = [int(v) for v in rng.integers(1, 301, 35)]
    cls = np.concatenate([np.full(k, c + 1, np.int32) for c, k in enumerate(per)])
    N = cls.shape[0]
    bx = synth_boxes(rng, N, 3.0, 800.0)
    sc = (rng.permutation(N) / N).astype(np.float32)
    keep = tb.nms_grouped(torch.from_numpy(bx).to(DEV), torch.from_numpy(sc).to(DEV), torch.from_numpy(cls).to(DEV), 0.5)
    ref = []
    for c in range(1, 36):
        idx = np.where(cls == c)[0]
        ref.append(idx[c_oracle.nms(np.ascontiguousarray(bx[idx]), np.ascontiguousarray(sc[idx]), 0.5, 0)])
    assert np.array_equal(keep.cpu().numpy(), np.sort(np.concatenate(ref)))


def test_config4_long_clip_100_objects_9900_pairs():
    """configs[4]: T = 64 (T' = 8), n = 100 -> 9 900 pairs; prd_k = 2 gives 19 800 candidates per clip, beyond one
    on-chip sort: ranked in segments on the device (no torch.topk fallback)."""
    from trace_amd.synthetic import make_inputs
    from tests.test_e2e_gpu import synth_labels_scores
    m = build()
    B, T, n = 1, 64, 100
    frames, blob, feat, rois = make_inputs(B, T, n, 224, seed=21)
    labels, scores = synth_labels_scores(rois, m.cfg.NUM_CLASSES, 13)
    sd = {k: v.detach().float().cpu() for k, v in m.state_dict().items()}
    m.to(DEV)
    outs = m.forward_clips(frames.to(DEV), blob.to(DEV), feat.to(DEV), rois, topk=100, prd_k=2, det_labels=labels,
                           det_scores=scores)
    assert m._debug["tmaps"].shape[1] == 8 and outs[0]["prd_scores"].shape[0] == 9900
    P = check_clip(m, sd, outs, 0, blob, feat, rois, 2, 0, labels, scores)
    assert P == 9900
