"""Multi-GPU path: clips shard across ranks (one process per GPU), no data-path collective; the only
exchange is ONE all-gather of fixed-size top-k triplet records per sweep, replacing the reference's
per-range pickle files merged by the parent process (lib/utils_rel/subprocess_rel.py:56-71,111-115;
lib/core/test_engine_rel.py:164-212).  Backend "nccl" is RCCL over xGMI on ROCm; "gloo" on CPU tests.
"""
import numpy as np
import torch
import torch.distributed as dist

RECORD_WIDTH = 14  # score, sbj, prd, obj, sbj box(4), obj box(4), clip id, valid count


def shard_range(n_items, rank, world):
    """Contiguous index range of `rank` — np.array_split semantics of subprocess_rel.py:57."""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def pack_triplets(outs, topk, clip_offset=0):
    """Per-clip return dicts (with top_* keys) -> (B, topk, RECORD_WIDTH) f32 tensor, zero padded.  The sweep's inner
    loop: no device -> host copy and no per-clip synchronisation - the host-side roi lists go up in one padded array and
    are gathered by the top pairs on the device."""
    B = len(outs)
    dev = outs[0]["prd_scores"].device if B else torch.device("cpu")
    rec = torch.zeros((B, topk, RECORD_WIDTH), dtype=torch.float32, device=dev)
    if B == 0:
        return rec
    ks = np.array([min(int(d["top_scores"].shape[0]), topk) for d in outs], np.int64)
    kmax = int(ks.max())
    pmax = max(int(d["sbj_rois"].shape[0]) for d in outs)
    boxes = np.zeros((B, max(pmax, 1), 8), np.float32)
    for b, d in enumerate(outs):
        boxes[b, :d["sbj_rois"].shape[0], :4] = d["sbj_rois"][:, 1:5]
        boxes[b, :d["obj_rois"].shape[0], 4:] = d["obj_rois"][:, 1:5]
    meta = np.zeros((B, topk, 3), np.float32)                       # [valid, clip id, count]
    meta[:, :, 0] = np.arange(topk)[None, :] < ks[:, None]
    meta[:, :, 1] = clip_offset + np.arange(B)[:, None]
    meta[:, :, 2] = ks[:, None]
    boxes_d = torch.from_numpy(boxes).to(dev, non_blocking=True)
    meta_d = torch.from_numpy(meta).to(dev, non_blocking=True)
    if kmax:
        pad = torch.nn.utils.rnn.pad_sequence

        def col(key):
            return pad([d[key][:topk] for d in outs], batch_first=True)
        rec[:, :kmax, 0] = col("top_scores")
        rec[:, :kmax, 1] = col("top_sbj").float()
        rec[:, :kmax, 2] = col("top_prd").float()
        rec[:, :kmax, 3] = col("top_obj").float()
        pair = col("top_pair").long().clamp_(0, boxes.shape[1] - 1)
        rec[:, :kmax, 4:12] = boxes_d.gather(1, pair[:, :, None].expand(-1, -1, 8))
        rec[:, :, :12] *= meta_d[:, :, :1]
    rec[:, :, 12:14] = meta_d[:, :, 1:]
    return rec


def gather_triplets(outs, topk, clip_offset=0):
    """All ranks' records, ordered by rank: (world*B, topk, RECORD_WIDTH).  Every rank must hold the
    same number of clips (pad the last shard with empty clips: trace_amd.sweep does)."""
    from .sweep import gather_records
    return gather_records(pack_triplets(outs, topk, clip_offset))
