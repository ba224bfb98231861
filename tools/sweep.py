#!/usr/bin/env python3
"""Sharded inference sweep (configs[3]): clip list -> shard_range -> steps (optionally overlapped) -> one all-gather of the
top-k triplet records -> rank-0 per-video ranking.  Synthetic clips of the bench workload's shape (no dataset offline).

    python tools/sweep.py --clips 64 --batch 8                                   one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        tools/sweep.py --clips 512 --batch 8                                      one rank per GPU (RCCL)
Prints one JSON line on rank 0: clips, videos, triplets kept, clips/s over the whole sweep (gather and ranking included).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=64)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--objects", type=int, default=20)
    ap.add_argument("--clips-per-video", type=int, default=4)
    ap.add_argument("--overlap", type=int, default=0, help="1: software-pipelined steps (faster; not bit-stable at bench size, DESIGN.md section 8)")
    args = ap.parse_args()
    import torch.distributed as dist
    from trace_amd.model_builder_rel import Generalized_RCNN
    from trace_amd.sweep import ModelStepper, rank_triplets_per_video, sweep
    from trace_amd.synthetic import make_inputs, randomize_model
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    torch.manual_seed(1)
    m = Generalized_RCNN()
    randomize_model(m, 1)
    m.eval().to(dev)
    m.graph_step = True
    m.overlap_steps = bool(args.overlap)
    # a small pool of resident synthetic batches, picked by clip index (stands in for the frame loader)
    pool = [tuple(t.to(dev) if torch.is_tensor(t) else t for t in make_inputs(args.batch, args.frames, args.objects, 224, seed=s))
            for s in range(2)]

    def source(ids):
        return pool[(ids[0] // args.batch) % len(pool)]

    stepper = ModelStepper(m, source)
    for _ in range(2):      # capture the graphs outside the timed sweep
        stepper([0] * args.batch, 0)
    stepper.flush()
    m.sync_steps()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    rec = sweep(stepper, args.clips, args.batch, 100, rank, world, flush_fn=stepper.flush, device=dev)
    vids = rank_triplets_per_video(rec, args.clips_per_video) if rank == 0 else None
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rank == 0:
        seen = sorted(set(int(c) for c in rec[:, 0, 12].cpu().tolist() if c >= 0))
        assert seen == list(range(args.clips)), "sweep lost clips"
        print(json.dumps({"clips": args.clips, "videos": len(vids), "triplets": int(sum(len(v) for v in vids.values())),
                          "n_gpus": world, "clips_per_sec": args.clips / dt, "seconds": dt}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
