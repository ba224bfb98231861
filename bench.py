#!/usr/bin/env python3
"""Headline benchmark: clips/sec of TRACE's per-clip relation hot path (I3D-R50 backbone ->
ROIAlign -> context aggregation + relation scoring) on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: starts N rank processes itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the hot path over one batch of synthetic clips per GPU: the workload
BASELINE.json's metric is quoted on (configs[1]: VidVRD res101xi3d50 setting, 32-frame 224^2 clips,
batch 8 per GPU, 20 objects / 380 pairs per clip, bf16).  Clips shard across ranks with no data-path
collective (weak scaling); the per-video top-100 triplet records are all-gathered (RCCL) once after
the timed region, as the sweep does.  Inputs are resident in HBM before timing.
Every step replays one captured HIP graph.  By default the steps run software-pipelined (--overlap 1,
model.overlap_steps: call n launches the backbone of step n beside the post-backbone chain of step n-1; the pipeline
is empty when the timed region starts and is flushed inside it) with the head's part A on a second stream beside the
backbone (--side-stream 1).  Both modes give the serial step's bits (tests/test_head_gpu.py::
test_step_graph_is_bit_stable_at_bench_size; DESIGN.md section 8 has the race that used to break this and its fix);
--overlap 0 --side-stream 0 times the serial step.  Triplets are ranked with prd_k = 20 predicates per pair, as the
reference's VidVRD evaluation does (lib/datasets_rel/task_evaluation_vg_and_vrd.py:95).

Prints ONE JSON line (rank 0) with the driver contract fields plus
  roofline     - dominant kernel (bf16 MFMA implicit-GEMM family): algorithmic FLOPs / launch time,
                 measured with events on the launch stream over the same K steps re-run launch by launch
                 right after the timed region (the timed steps replay the backbone from a HIP graph)
  cpu_baseline - the CPU oracle (a port of the reference path) timed on the host cores (rank 0, N=1)
  roofline_roi_align* / roofline_pair / roofline_nms - the bandwidth kernels of the path against the HBM roofline
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK = 2.5e15  # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK = 8.0e12        # HBM3E, same guide
I3D_GFLOP_PER_CLIP = 108.1  # reference algorithm @ 32 x 224^2


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="clips per GPU per step")
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--objects", type=int, default=20)
    ap.add_argument("--pipeline", type=int, default=1,
                    help="captured step graphs in flight per GPU (2 measured no faster than 1: 2634 vs 2653 clips/s)")
    ap.add_argument("--overlap", type=int, default=1,
                    help="1: a step's post-backbone chain runs beside the next step's backbone (model.overlap_steps)")
    ap.add_argument("--split-stage", type=int, default=2,
                    help="with --overlap: cut the pipelined step behind this residual stage (0-based; 2 = layer3) instead of "
                         "behind the backbone (-1): stem .. layer3 + part A of step n then run beside layer4 + the head of "
                         "step n-1 (measured 2.46 -> 2.34 ms per step; same bits)")
    ap.add_argument("--split-block", type=int, default=3,
                    help="> 0: cut behind the first k blocks of stage --split-stage instead of behind the whole stage (0); "
                         "measured behind layer3's blocks 1 / 2 / 3 / 4 / 5 / 6: 2.39 / 2.28 / 2.25-2.27 / 2.31 / 2.31 / 2.34 ms")
    ap.add_argument("--side-fork", type=int, default=None,
                    help="residual stage behind which part A forks onto the second stream (model.side_fork_stage; -1 = stem)")
    ap.add_argument("--side-stream", type=int, default=1,
                    help="1: the head's part A runs on a second stream beside the backbone")
    ap.add_argument("--prd-k", type=int, default=20, help="predicates ranked per pair (reference: prd_k_set = (20,))")
    ap.add_argument("--no-secondary", action="store_true", help="skip the configs[2] / configs[4] lines")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-trace", action="store_true", help="skip per-launch events (roofline leg)")
    return ap.parse_args()


def cpu_baseline(model_sd, frames, blob, feat, rois, budget_s=25.0, min_s=12.0):
    """CPU oracle ("port" of the reference path: oracle/i3d_ref.py + oracle/head_ref.py, fp32) on a
    bounded sample of the same workload: whole clips, one at a time as the reference does, cycling through the step's
    clips until at least `min_s` seconds of CPU work are done (and stopping at `budget_s`)."""
    from oracle.i3d_ref import i3d_forward
    from oracle.head_ref import head_forward
    torch.set_num_threads(min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16"))))
    cores = torch.get_num_threads()
    i3d_sd = {k[len("Prd_RCNN."):]: v for k, v in model_sd.items() if k.startswith("Prd_RCNN.")}
    done, t0 = 0, time.time()
    with torch.no_grad():
        while done == 0 or (time.time() - t0 < budget_s and (done < frames.shape[0] or time.time() - t0 < min_s)):
            b = done % frames.shape[0]
            tm = i3d_forward(i3d_sd, frames[b:b + 1].permute(0, 2, 1, 3, 4).contiguous())   # (1,C,T',h,w)
            head_forward(model_sd, blob[b:b + 1], feat[b:b + 1], tm[0].permute(1, 0, 2, 3).contiguous(), rois[b])
            done += 1
    dt = time.time() - t0
    return dict(value=done / dt, unit="clips/sec", cores=cores, kind="port",
                sample="%d clip(s) of the bench workload (%df 224^2, %d objects), fp32, one clip per call"
                       % (done, frames.shape[1], rois[0].shape[0]))


def secondary_configs(dev, prd_k, steps=5, only=None, kernels=True):
    """BASELINE configs[2] (Action Genome setting: 50 objects, all ordered pairs through the overlap filter, AG dims) and
    configs[4] (64 frames, 100 objects, 9 900 pairs) as secondary lines: whole steps from the captured graph (second
    stream on), plus the ROIAlign / pair-kernel rooflines at those sizes.  configs[1] stays the headline."""
    import bench_kernels
    from trace_amd.config import ag_config
    from trace_amd.model_builder_rel import Generalized_RCNN
    from trace_amd.synthetic import make_inputs, randomize_model
    out = []
    for tag, cfg, B, T, n in (("configs[2]: Action Genome setting, 32-frame clips, 50 objects, overlap filter", ag_config(), 4, 32, 50),
                              ("configs[4]: 64-frame clip, 100 objects", None, 1, 64, 100)):
        if only is not None and only not in tag:
            continue
        torch.manual_seed(2)
        m = Generalized_RCNN(cfg)
        randomize_model(m, 2)
        m.eval().to(dev)
        calib = torch.randn((1, 3, 16, 112, 112), generator=torch.Generator().manual_seed(1)) * 50
        m.Prd_RCNN.calibrate_bn(calib.to(dev))
        fr, bl, fe, rois = make_inputs(B, T, n, 224, seed=3)
        fr = fr.to(dev)
        bl = bl.to(dev).permute(0, 2, 3, 1).contiguous().to(torch.bfloat16)
        fe = fe.to(dev).permute(0, 2, 3, 1).contiguous().to(torch.bfloat16)
        rng = np.random.default_rng(5)
        labels = [rng.integers(2, m.cfg.NUM_CLASSES, n).astype(np.int32) for _ in range(B)]     # no persons: every ordered pair
        scores = [(rng.permutation(n) / n * 0.9 + 0.05).astype(np.float32) for _ in range(B)]
        m.graph_step, m.side_stream = True, True
        kk = min(prd_k, m.cfg.NUM_PRD_CLASSES)

        def step():
            return m.forward_clips(fr, bl, fe, rois, topk=100, prd_k=kk, det_labels=labels, det_scores=scores)
        for _ in range(2):
            outs = step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            outs = step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        pairs = [int(o["prd_scores"].shape[0]) for o in outs]
        if not kernels:
            out.append(dict(workload=tag, value=B / dt, ms_per_step=dt * 1e3, pairs_per_clip=pairs))
            continue
        kern = bench_kernels.measure(dev, B=B, n=n, Tn=m.Prd_RCNN.out_frames(T), C=int(bl.shape[3]), fm=int(bl.shape[1]), nms=False)
        out.append(dict(workload=tag, clips_per_gpu=B, frames=T, objects=n, pairs_per_clip=pairs, prd_k=kk,
                        value=B / dt, unit="clips/sec", ms_per_step=dt * 1e3, pairs_per_sec=sum(pairs) / dt,
                        roofline_roi_align=kern["roi_align"], roofline_roi_align_avg=kern["roi_align_avg"],
                        roofline_pair={"pool": kern["pair_pool"], "combine": kern["pair_combine"], "fused": kern["pair_infer"], "mfma": kern["pair_mfma"]}))
        m.drop_graphs()
        del m
        torch.cuda.empty_cache()
    return out


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N rank processes (one per GPU, torch.distributed.run on
    127.0.0.1) as CHILDREN of this process - which has not touched the GPU - and exit with their status."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd))


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE): refusing to report a "
                         "number for the wrong GPU count" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist_mod.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        dist = dist_mod
        seen = torch.ones((1,), device=dev)
        dist.all_reduce(seen)                      # RCCL itself must see every rank
        if int(seen.item()) != args.gpus:
            raise SystemExit("bench.py: RCCL sees %d ranks, --gpus says %d" % (int(seen.item()), args.gpus))

    from trace_amd import ops
    from trace_amd.model_builder_rel import Generalized_RCNN
    from trace_amd.synthetic import make_inputs, randomize_model
    from trace_amd.dist import gather_triplets

    torch.manual_seed(1)
    model = Generalized_RCNN()
    randomize_model(model, 1)
    model.eval()
    sd_cpu = None
    model.to(dev)
    model.graph_step = True     # serving mode: the whole step replays from one HIP graph per step signature
    # Both concurrency modes of the model on by default (part A on a second stream beside the backbone; the previous
    # step's tail beside the next backbone): 3200 vs 2800 clips/s, same bits as the serial step.
    model.side_stream = bool(args.side_stream)
    model.pipeline = args.pipeline   # step graphs in flight (2: the head of step i overlaps the backbone of step i+1)
    # software-pipelined step graphs: call n runs the backbone of step n beside the rest of step n-1 and returns the
    # results of step n-1; every timed region below ends with flush_steps(), so K calls = K complete steps
    model.overlap_steps = bool(args.overlap) and args.pipeline == 1
    model.split_stage = args.split_stage if args.split_stage >= 0 else None
    if model.split_stage is not None and args.split_block > 0:
        model.split_stage = (args.split_stage, args.split_block)
    if args.side_fork is not None:
        model.side_fork_stage = args.side_fork
    calib = torch.randn((1, 3, 16, 112, 112), generator=torch.Generator().manual_seed(1)) * 50
    model.Prd_RCNN.calibrate_bn(calib.to(dev))       # setup only: sane activations for random weights
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sd_cpu = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}

    B, T, n = args.batch, args.frames, args.objects
    frames_c, blob_c, feat_c, rois = make_inputs(B, T, n, 224, seed=1 + rank)
    frames = frames_c.to(dev)
    blob = blob_c.to(dev).permute(0, 2, 3, 1).contiguous().to(torch.bfloat16)    # Conv_Body output, channels-last
    feat = feat_c.to(dev).permute(0, 2, 3, 1).contiguous().to(torch.bfloat16)    # Prd_RCNN2D output

    state = dict(resident=False)

    def step():
        # the inputs stay resident in the buffers the captured step graph reads (no per-step copy); with two
        # graph instances in flight each has its own copy of the (identical) synthetic inputs
        f, b, m_ = model.step_buffers() if state["resident"] else (frames, blob, feat)
        return model.forward_clips(f, b, m_, rois, topk=100, prd_k=args.prd_k)

    for _ in range(max(args.warmup, args.pipeline)):
        outs = step()
    state["resident"] = True
    for _ in range(args.pipeline):
        outs = step()
    if model.overlap_steps:
        outs = model.flush_steps()
    model.sync_steps()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        outs = step()
    if model.overlap_steps:
        outs = model.flush_steps()        # the last step's tail: inside the timed region
    model.sync_steps()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    # roofline leg: the same K steps with every igemm launch bracketed by events on the launch stream
    # (launch by launch, i.e. without the graph replay, whose nodes cannot carry events)
    trace = None
    if rank == 0 and not args.no_trace:
        trace = []
        ops.IGEMM_TRACE = trace
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        ops.IGEMM_TRACE = None
        # what an (event, launch, event) bracket costs beyond the kernel itself: empty brackets on the same stream
        empty = []
        for _ in range(200):
            s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s0.record()
            e0.record()
            empty.append((s0, e0))
        torch.cuda.synchronize()
        event_overhead_ms = float(np.median([a.elapsed_time(b) for a, b in empty]))
    if dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # the sweep's single exchange step: all-gather the fixed-size top-100 records of every clip
    rec = gather_triplets(outs, 100, clip_offset=rank * B)
    assert rec.shape[0] == world * B

    if rank != 0:
        if dist:
            dist.destroy_process_group()
        return
    value = world * B * args.steps / dt
    roof = None
    if trace:
        ms = [t[0].elapsed_time(t[1]) for t in trace]   # as the events give it (an empty bracket, reported beside it, is NOT subtracted)
        fl = [t[2] for t in trace]
        alg_bytes = sum(t[3] for t in trace) / len(trace)
        # HBM bytes per launch from the PMC counters (FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 passes over
        # this same command, tools/pmc_traffic.py): bench.py cannot run under the profiler itself, so it quotes
        # the committed summary of the current round
        traffic = None
        for tp in ("r03_igemm_traffic.json", "r02_igemm_traffic.json", "r01_igemm_traffic.json"):
            tp = os.path.join(ROOT, "profiles", tp)
            if os.path.exists(tp):
                traffic = json.load(open(tp)).get("traffic_bytes_per_launch")
                break
        avg_ms = sum(ms) / len(ms)
        achieved = (sum(fl) / len(fl)) / (avg_ms * 1e-3)      # avg algorithmic FLOPs per launch / avg launch duration
        roof = dict(bound="mfma", achieved=achieved / 1e12, peak=MFMA_BF16_PEAK / 1e12, unit="TFLOP/s",
                    frac=achieved / MFMA_BF16_PEAK, traffic=traffic, traffic_unit="bytes per launch (PMC)",
                    algorithmic_bytes_per_launch=alg_bytes, kernel="trace::igemm*_bf16_kernel<*>",
                    launches_per_step=len(trace) // args.steps, avg_launch_us=avg_ms * 1e3,
                    event_bracket_overhead_us=event_overhead_ms * 1e3,
                    gflop_per_step=sum(fl) / args.steps / 1e9)
    # The bandwidth-bound kernels of the path against the HBM roofline, each timed alone at this
    # workload's shapes with events around a captured graph of 20 launches (tools/bench_kernels.py): ROIAlign (object
    # heads; adaptive grid; the fused mean variant), pair pooling / combine, NMS at the stress sizes.
    # Algorithmic bytes = outputs written once + maps read once + rois.
    kern = None
    if trace is not None:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_kernels
        kern = bench_kernels.measure(dev, B=B, n=n, C=int(blob.shape[3]), fm=int(blob.shape[1]))
    # PCIe-inclusive rate (never `value`): every step first copies its clip from pinned host memory into the resident
    # buffer on the step's stream, un-overlapped - as the reference's f32 blob (test_engine_rel.py:420-520) and as
    # decoded uint8 frames (trace_conv3d_stem16_u8: a quarter of the bytes)
    pcie = None
    if rank == 0 and not args.no_trace:
        pcie = {}
        model.Prd_RCNN.pixel_means = [102.9801, 115.9465, 122.7717]
        for tag, host in (("f32", torch.empty(tuple(frames.shape), dtype=torch.float32).normal_(0, 50).pin_memory()),
                          ("u8", torch.randint(0, 256, (B, T, 224, 224, 3), dtype=torch.uint8).pin_memory())):
            res = torch.empty(tuple(host.shape), dtype=host.dtype, device=dev)
            model.overlap_steps = 0

            def pstep():
                res.copy_(host, non_blocking=True)
                return model.forward_clips(res, blob, feat, rois, topk=100, prd_k=args.prd_k)
            for _ in range(3):
                pstep()
            model.sync_steps()
            torch.cuda.synchronize()
            tp0 = time.perf_counter()
            for _ in range(args.steps):
                pstep()
            model.sync_steps()
            torch.cuda.synchronize()
            pcie[tag] = dict(clips_per_sec=B * args.steps / (time.perf_counter() - tp0), host_bytes_per_step=host.numel() * host.element_size())
    second = None
    if trace is not None and not args.no_secondary:
        model.drop_graphs()
        second = secondary_configs(dev, args.prd_k)
    cpu = None
    if sd_cpu is not None:
        cpu = cpu_baseline(sd_cpu, frames_c, blob_c, feat_c, rois)
    out = {
        "metric": "clips/sec (32f 224^2, I3D-R50 + TRACE head)",
        "value": value, "unit": "clips/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "VidVRD res101xi3d50 setting (configs[1]): %d-frame 224^2 clips, batch %d per GPU, "
                               "%d objects (%d pairs) per clip, I3D-R50-DC5 + ROIAlign + TRACE relation head; "
                               "2-D conv-body outputs synthetic" % (T, B, n, n * (n - 1)),
                   "clips_per_gpu": B, "frames": T, "objects": n, "parallelism": "clip-sharded x%d" % world},
        "roofline": roof, "cpu_baseline": cpu, "pcie_inclusive": pcie,
        "roofline_roi_align": kern and kern["roi_align"],
        "roofline_roi_align_adaptive": kern and kern["roi_align_adaptive"],
        "roofline_roi_align_avg": kern and kern["roi_align_avg"],
        "roofline_pair": kern and {"pool": kern["pair_pool"], "combine": kern["pair_combine"], "fused": kern["pair_infer"], "mfma": kern["pair_mfma"]},
        "roofline_nms": kern and {"rpn": kern["nms_rpn"], "per_class": kern["nms_per_class"]},
        "i3d_algorithmic_tflops": value * I3D_GFLOP_PER_CLIP / 1e3,
        "prd_k": args.prd_k, "overlap_steps": bool(args.overlap) and args.pipeline == 1, "side_stream": bool(args.side_stream),
        "split_stage": ((args.split_stage, args.split_block) if args.split_block > 0 else args.split_stage) if (bool(args.overlap) and args.pipeline == 1 and args.split_stage >= 0) else None,
        "other_configs": second,
    }
    print(json.dumps(out))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
