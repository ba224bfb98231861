"""N>1 path on CPU: world_size-2 gloo run of the clip sharding + triplet all-gather."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fake_outs(rank, B, topk):
    outs = []
    g = torch.Generator().manual_seed(100 + rank)
    for b in range(B):
        k = topk - b  # ragged valid counts
        P = 30
        rois = np.concatenate([np.zeros((P, 1), np.float32), np.random.default_rng(rank * 10 + b).uniform(0, 200, (P, 4)).astype(np.float32)], 1)
        outs.append(dict(prd_scores=torch.rand((P, 5), generator=g), top_scores=torch.rand((k,), generator=g),
                         top_pair=torch.randint(0, P, (k,), generator=g), top_prd=torch.randint(0, 5, (k,), generator=g),
                         top_sbj=torch.randint(1, 36, (k,), generator=g), top_obj=torch.randint(1, 36, (k,), generator=g),
                         sbj_rois=rois, obj_rois=rois[::-1].copy()))
    return outs


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from trace_amd.dist import gather_triplets, pack_triplets, shard_range
    B, topk = 3, 10
    outs = _fake_outs(rank, B, topk)
    rec = gather_triplets(outs, topk, clip_offset=rank * B)
    mine = pack_triplets(outs, topk, clip_offset=rank * B)
    ok = rec.shape == (world * B, topk, 14) and torch.equal(rec[rank * B:(rank + 1) * B], mine)
    other = pack_triplets(_fake_outs(1 - rank, B, topk), topk, clip_offset=(1 - rank) * B)
    ok = ok and torch.equal(rec[(1 - rank) * B:(2 - rank) * B], other)
    ok = ok and rec[:, 0, 12].tolist() == list(range(world * B))
    q.put((rank, bool(ok), shard_range(7, rank, world)))
    dist.destroy_process_group()


def test_two_rank_gloo_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
    assert [r[1] for r in res] == [True, True]
    assert res[0][2] == (0, 4) and res[1][2] == (4, 7)      # np.array_split(range(7), 2)


def test_shard_range_matches_array_split():
    from trace_amd.dist import shard_range
    for n in (0, 1, 7, 8, 100):
        for w in (1, 2, 3, 8):
            parts = np.array_split(np.arange(n), w)
            for r in range(w):
                s, e = shard_range(n, r, w)
                assert list(range(s, e)) == parts[r].tolist()


def _sweep_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from trace_amd.sweep import n_batches_per_rank, rank_triplets_per_video, sweep
    n_clips, batch, topk = 11, 4, 6          # shards 6 + 5: ragged last batches, one of them shorter than the other rank's
    calls = []

    class Lagged:                             # a software-pipelined stepper: a call returns the previous batch
        def __init__(self):
            self.prev = None

        def make(self, ids):
            outs = []
            for c in ids:
                g = torch.Generator().manual_seed(1000 + c)
                k = 1 + c % topk
                P = 12
                rois = np.concatenate([np.zeros((P, 1), np.float32), np.random.default_rng(c).uniform(0, 200, (P, 4)).astype(np.float32)], 1)
                outs.append(dict(prd_scores=torch.rand((P, 5), generator=g), top_scores=torch.sort(torch.rand((k,), generator=g), descending=True)[0],
                                 top_pair=torch.randint(0, P, (k,), generator=g), top_prd=torch.randint(0, 5, (k,), generator=g),
                                 top_sbj=torch.randint(1, 36, (k,), generator=g), top_obj=torch.randint(1, 36, (k,), generator=g),
                                 sbj_rois=rois, obj_rois=rois[::-1].copy()))
            return outs

        def __call__(self, ids, n_pad):
            calls.append((list(ids), n_pad))
            prev, self.prev = self.prev, ids
            return None if prev is None else (prev, self.make(prev) + [None] * 3)   # trailing padding outputs are ignored

        def flush(self):
            ids, self.prev = self.prev, None
            return ids, self.make(ids)

    st = Lagged()
    rec = sweep(st, n_clips, batch, topk, rank, world, flush_fn=st.flush)
    nb = n_batches_per_rank(n_clips, world, batch)
    ok = nb == 2 and rec.shape == (world * nb * batch, topk, 14)
    ids = [int(c) for c in rec[:, 0, 12].tolist()]
    ok = ok and sorted(c for c in ids if c >= 0) == list(range(n_clips)) and ids.count(-1) == world * nb * batch - n_clips
    # rank 0 holds clips 0-5 (batches 4 + 2), rank 1 clips 6-10 (batches 4 + 1)
    ok = ok and calls == ([([0, 1, 2, 3], 0), ([4, 5], 2)] if rank == 0 else [([6, 7, 8, 9], 0), ([10], 3)])
    vids = rank_triplets_per_video(rec, clips_per_video=4, top=9)
    ok = ok and sorted(vids) == [0, 1, 2]
    for v, rows in vids.items():
        ok = ok and len(rows) <= 9 and bool(np.all(np.diff(rows[:, 0]) <= 0)) and set((rows[:, 12] // 4).astype(int)) == {v}
    # the records of a clip are the ones its owner packed
    from trace_amd.dist import pack_triplets
    want = pack_triplets(st.make([7]), topk, clip_offset=7)[0]
    ok = ok and torch.equal(rec[ids.index(7)], want)
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_two_rank_gloo_sweep_with_uneven_shards():
    """tools/sweep.py's loop on CPU: 11 clips over 2 ranks in batches of 4 (shards of 6 and 5 clips, ragged last
    batches), a stepper with the one-call lag of overlapped steps, one all-gather, rank-0 style per-video ranking."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    ps = [ctx.Process(target=_sweep_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
    assert [r[1] for r in res] == [True, True]


def test_pack_triplets_matches_the_per_clip_loop():
    """The vectorised packer against the plain per-clip loop it replaced: ragged k, ragged pair counts, clip offset."""
    import numpy as np
    import torch
    from trace_amd.dist import RECORD_WIDTH, pack_triplets
    rng = np.random.default_rng(3)
    outs, topk = [], 7
    for k, P in ((7, 12), (3, 4), (0, 2), (5, 30)):
        pair = torch.from_numpy(rng.integers(0, P, k))
        outs.append(dict(prd_scores=torch.zeros((P, 3)), top_scores=torch.from_numpy(rng.random(k).astype(np.float32)),
                         top_sbj=torch.from_numpy(rng.integers(1, 36, k)), top_prd=torch.from_numpy(rng.integers(0, 132, k)),
                         top_obj=torch.from_numpy(rng.integers(1, 36, k)), top_pair=pair,
                         sbj_rois=rng.random((P, 5)).astype(np.float32), obj_rois=rng.random((P, 5)).astype(np.float32)))
    want = torch.zeros((len(outs), topk, RECORD_WIDTH))
    for b, d in enumerate(outs):
        k = int(d["top_scores"].shape[0])
        pair = d["top_pair"].numpy()
        want[b, :k, 0] = d["top_scores"]
        want[b, :k, 1], want[b, :k, 2], want[b, :k, 3] = d["top_sbj"].float(), d["top_prd"].float(), d["top_obj"].float()
        want[b, :k, 4:8] = torch.from_numpy(d["sbj_rois"][pair, 1:5])
        want[b, :k, 8:12] = torch.from_numpy(d["obj_rois"][pair, 1:5])
        want[b, :, 12] = 40 + b
        want[b, :, 13] = k
    assert torch.equal(pack_triplets(outs, topk, clip_offset=40), want)
