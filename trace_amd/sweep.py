"""Inference sweep over a clip list, sharded by clip across ranks (configs[3]).

The reference starts one subprocess per GPU with `--range start end` (np.array_split of the clip list), each writes a
pickle of its range and the parent concatenates them (lib/utils_rel/subprocess_rel.py:42-116,
lib/core/test_engine_rel.py:164-212).  Here every rank (one process per GPU, torch.distributed) walks its own
contiguous range in fixed-size batches - the step graph wants one shape - keeps the fixed-size top-k records of its
clips on the device, and the whole sweep has ONE exchange: an all-gather of those records at the end.  Rank 0 drops
the padding and ranks the triplets per video.
"""
import numpy as np
import torch
import torch.distributed as dist

from .dist import RECORD_WIDTH, pack_triplets, shard_range


def n_batches_per_rank(n_clips, world, batch):
    """Every rank runs the same number of batches (the all-gather needs equal sizes): that of the longest shard,
    which np.array_split gives to rank 0."""
    s, e = shard_range(n_clips, 0, world)
    return (e - s + batch - 1) // batch


def gather_records(rec):
    """(R, topk, RECORD_WIDTH) per rank -> (world*R, topk, RECORD_WIDTH), ordered by rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return rec
    world = dist.get_world_size()
    out = torch.empty((world * rec.shape[0],) + tuple(rec.shape[1:]), dtype=rec.dtype, device=rec.device)
    dist.all_gather_into_tensor(out, rec.contiguous())
    return out


def sweep(step_fn, n_clips, batch, topk, rank, world, flush_fn=None, device="cpu"):
    """Run `step_fn` over this rank's shard of range(n_clips) and gather every rank's records.

    step_fn(ids, n_pad): ids = the consecutive clip indices of one batch (1 <= len(ids) <= batch), n_pad = batch -
      len(ids) clips of padding the callee must add to keep the step's shape.  Returns None or (ids_done, outs):
      the per-clip return dicts (model.forward_clips) of a batch - this one or, for a software-pipelined stepper,
      an earlier one.  flush_fn() returns the same for whatever is still in flight.
    Returns (world * n_batches * batch, topk, RECORD_WIDTH) f32 records ordered by rank; padding has clip id -1.
    """
    start, end = shard_range(n_clips, rank, world)
    nb = n_batches_per_rank(n_clips, world, batch)
    rec = torch.zeros((nb * batch, topk, RECORD_WIDTH), dtype=torch.float32, device=device)
    rec[:, :, 12] = -1.0
    slot_of = {}

    def take(done):
        if done is None:
            return
        ids, outs = done
        if len(ids):
            r = pack_triplets(outs[:len(ids)], topk, clip_offset=ids[0]).to(rec.device)
            s = slot_of.pop(ids[0])
            rec[s:s + len(ids)] = r

    for i in range(nb):
        ids = list(range(start + i * batch, min(start + (i + 1) * batch, end)))
        if not ids:
            continue            # a shorter shard: this batch is padding only
        slot_of[ids[0]] = i * batch
        take(step_fn(ids, batch - len(ids)))
    if flush_fn is not None:
        take(flush_fn())
    if slot_of:
        raise RuntimeError("sweep: the stepper never returned clips %s" % sorted(slot_of))
    return gather_records(rec)


def rank_triplets_per_video(records, clips_per_video=1, top=200):
    """Rank 0's merge (tools/transform_vidvrd_results.py:154-265 takes per-frame triplet lists per video): drop the
    padding, order by clip, group clips into videos and sort every video's triplets by score, descending, ties by
    (clip, position) - a stable order.  Returns {video: (k, RECORD_WIDTH) f32 array}, k <= top."""
    r = records.detach().cpu().numpy().reshape(-1, records.shape[1], RECORD_WIDTH)
    out = {}
    clip_ids = r[:, 0, 12].astype(np.int64)
    for ci in np.argsort(clip_ids, kind="stable"):
        c = int(clip_ids[ci])
        if c < 0:
            continue
        k = int(r[ci, 0, 13])
        out.setdefault(c // clips_per_video, []).append(r[ci, :k])
    for v in list(out):
        rows = np.concatenate(out[v], 0) if out[v] else np.zeros((0, RECORD_WIDTH), np.float32)
        order = np.argsort(-rows[:, 0], kind="stable")[:top]
        out[v] = rows[order]
    return out


class ModelStepper:
    """step_fn / flush_fn over a Generalized_RCNN: pads a short batch by repeating its last clip, and follows the
    one-call lag of model.overlap_steps (a call returns the previous call's clips)."""

    def __init__(self, model, clip_source, topk=100, prd_k=1):
        self.model, self.src, self.topk, self.prd_k = model, clip_source, topk, prd_k
        self.inflight = None

    def __call__(self, ids, n_pad):
        frames, blob, feat, rois = self.src(ids + [ids[-1]] * n_pad)
        outs = self.model.forward_clips(frames, blob, feat, rois, topk=self.topk, prd_k=self.prd_k)
        if not self.model.overlap_steps:
            return ids, outs
        prev, self.inflight = self.inflight, ids
        return None if prev is None or outs is None else (prev, outs)

    def flush(self):
        if not self.model.overlap_steps or self.inflight is None:
            return None
        ids, self.inflight = self.inflight, None
        return ids, self.model.flush_steps()
