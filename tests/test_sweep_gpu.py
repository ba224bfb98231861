"""BASELINE configs[3] on one GPU: the sharded inference sweep (trace_amd.sweep.sweep + ModelStepper, world = 1) over 11
clips in batches of 4 - a ragged last batch whose padding is dropped - must hand back, for every clip, exactly the
top-k triplet records of that clip's own step (model.forward_clips on the same batch, launched eagerly), in the plain
step graph and in the software-pipelined mode with its one-call lag.  The N > 1 exchange (one all-gather) is covered by
the 2-rank gloo tests in tests/test_dist_cpu.py; reference: lib/utils_rel/subprocess_rel.py:42-116."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_sweep_reproduces_per_clip_topk_ragged_last_batch():
    from trace_amd.dist import pack_triplets
    from trace_amd.model_builder_rel import Generalized_RCNN
    from trace_amd.sweep import ModelStepper, rank_triplets_per_video, sweep
    from trace_amd.synthetic import make_inputs, randomize_model
    torch.manual_seed(1)
    m = Generalized_RCNN()
    randomize_model(m, 1)
    m.eval().to(DEV)
    m.Prd_RCNN.calibrate_bn((torch.randn((1, 3, 8, 64, 64), generator=torch.Generator().manual_seed(1)) * 50).to(DEV))
    n_clips, batch, topk = 11, 4, 50
    fr, bl, fe, rois = make_inputs(n_clips, 8, 6, 224, seed=9)
    fr, bl, fe = fr.to(DEV), bl.to(DEV), fe.to(DEV)
    calls = []

    def source(ids):
        calls.append(list(ids))
        ix = torch.tensor(ids, device=DEV)
        return fr[ix].contiguous(), bl[ix].contiguous(), fe[ix].contiguous(), [rois[i] for i in ids]

    # reference: every batch as the stepper composes it (short batch padded by repeating its last clip), eager launches
    want = torch.zeros((12, topk, 14), device=DEV)
    want[:, :, 12] = -1
    for s in range(0, n_clips, batch):
        ids = list(range(s, min(s + batch, n_clips)))
        padded = ids + [ids[-1]] * (batch - len(ids))
        outs = m.forward_clips(*source(padded), topk=topk, prd_k=2)
        want[s:s + len(ids)] = pack_triplets(outs[:len(ids)], topk, clip_offset=s)
    torch.cuda.synchronize()
    assert (want[:n_clips, 0, 13] > 0).all() and float(want[n_clips, 0, 12]) == -1.0
    try:
        for overlap in (False, True):
            m.graph_step, m.overlap_steps = True, overlap
            st = ModelStepper(m, source, topk=topk, prd_k=2)
            calls.clear()
            rec = sweep(st, n_clips, batch, topk, 0, 1, flush_fn=st.flush, device=DEV)
            torch.cuda.synchronize()
            assert calls == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10, 10]]
            assert rec.shape == (12, topk, 14)
            assert torch.equal(rec, want), "overlap %s: records differ from the per-batch eager steps" % overlap
            vids = rank_triplets_per_video(rec, clips_per_video=4, top=100)
            assert sorted(vids) == [0, 1, 2]
            assert all(np.all(np.diff(v[:, 0]) <= 0) for v in vids.values())
            assert set(int(c) for v in vids.values() for c in v[:, 12]) == set(range(n_clips))   # the padding is gone
            m.drop_graphs()
    finally:
        m.overlap_steps = False
        m.graph_step = False
        m.drop_graphs()
