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
    """Per-clip return dicts (with top_* keys) -> (B, topk, RECORD_WIDTH) f32 tensor, zero padded."""
    dev = outs[0]["prd_scores"].device if len(outs) else torch.device("cpu")
    rec = torch.zeros((len(outs), topk, RECORD_WIDTH), dtype=torch.float32, device=dev)
    for b, d in enumerate(outs):
        k = int(d["top_scores"].shape[0])
        pair = d["top_pair"].cpu().numpy()
        rec[b, :k, 0] = d["top_scores"]
        rec[b, :k, 1] = d["top_sbj"].float()
        rec[b, :k, 2] = d["top_prd"].float()
        rec[b, :k, 3] = d["top_obj"].float()
        rec[b, :k, 4:8] = torch.from_numpy(np.ascontiguousarray(d["sbj_rois"][pair, 1:5])).to(dev)
        rec[b, :k, 8:12] = torch.from_numpy(np.ascontiguousarray(d["obj_rois"][pair, 1:5])).to(dev)
        rec[b, :, 12] = clip_offset + b
        rec[b, :, 13] = k
    return rec


def gather_triplets(outs, topk, clip_offset=0):
    """All ranks' records, ordered by rank: (world*B, topk, RECORD_WIDTH).  Every rank must hold the
    same number of clips (pad the last shard with empty clips: trace_amd.sweep does)."""
    from .sweep import gather_records
    return gather_records(pack_triplets(outs, topk, clip_offset))
