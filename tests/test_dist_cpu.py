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
