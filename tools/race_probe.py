#!/usr/bin/env python3
"""Race screen at the bench size (8 clips x 32 frames, 20 objects): replays the captured step in one of the model's
modes and reports every replay whose results differ from the first one, with the first backbone block whose output
differs and where (clip, rows, channels) - so that one bad replay names the launch that went wrong.

    python tools/race_probe.py --mode side --reps 40 [--alternate] --out gpurun_out/race_side.json
    TRACE_AMD_LIB=trace_amd/lib/libtrace_amd_<variant>.so python tools/race_probe.py ...
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="side", choices=["serial", "side", "overlap", "overlap_side"])
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--alternate", action="store_true", help="alternate two input sets (stale data becomes visible)")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--objects", type=int, default=20)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    from trace_amd import ops
    from trace_amd.model_builder_rel import Generalized_RCNN
    from trace_amd.synthetic import make_inputs, randomize_model
    dev = torch.device("cuda", 0)
    torch.manual_seed(1)
    m = Generalized_RCNN()
    randomize_model(m, 1)
    m.eval().to(dev)
    calib = torch.randn((1, 3, 16, 112, 112), generator=torch.Generator().manual_seed(1)) * 50
    m.Prd_RCNN.calibrate_bn(calib.to(dev))
    sets = []
    for seed in (31, 32):
        fr, bl, fe, ro = make_inputs(args.batch, args.frames, args.objects, 224, seed=seed)
        sets.append((fr.to(dev), bl.to(dev), fe.to(dev), ro))
    keys = ("prd_scores", "msp_det_scores", "top_scores", "top_pair", "top_prd")
    m.graph_step = True
    m.side_stream = args.mode in ("side", "overlap_side")
    m.overlap_steps = args.mode in ("overlap", "overlap_side")
    probing = not m.overlap_steps
    if probing:
        ops.PROBE = []

    def run(i):
        c = sets[i]
        outs = m.forward_clips(c[0], c[1], c[2], c[3], topk=100, prd_k=1)
        return outs

    def snap(outs):
        torch.cuda.synchronize()
        return [{k: o[k].clone() for k in keys} for o in outs]

    order = [(r % 2) if args.alternate else 0 for r in range(args.reps)]
    report = dict(mode=args.mode, reps=args.reps, alternate=bool(args.alternate), lib=os.environ.get("TRACE_AMD_LIB", "default"),
                  bad=[])
    if m.overlap_steps:
        got = []
        for i in order:
            outs = run(i)
            if outs is not None:
                got.append(snap(outs))
        got.append(snap(m.flush_steps()))
        ref = {}
        for r, (i, g) in enumerate(zip(order, got)):
            if i not in ref:
                ref[i] = g
                continue
            clips = [ci for ci, (a, b) in enumerate(zip(g, ref[i])) if any(not torch.equal(a[k], b[k]) for k in keys)]
            if clips:
                report["bad"].append(dict(rep=r, clips=clips))
    else:
        first = {}
        blocks = None
        for r, i in enumerate(order):
            outs = snap(run(i))
            if blocks is None:
                n = len(ops.PROBE) // 2          # warm-up pass + captured pass: the graph's buffers are the last half
                blocks = ops.PROBE[n:]
                ops.PROBE = None
            pr = [(name, t.clone()) for name, t in blocks]
            torch.cuda.synchronize()
            if i not in first:
                first[i] = (outs, pr)
                continue
            ro, rp = first[i]
            clips = [ci for ci, (a, b) in enumerate(zip(outs, ro)) if any(not torch.equal(a[k], b[k]) for k in keys)]
            where = None
            for (name, t), (_, t0) in zip(pr, rp):
                if not torch.equal(t, t0):
                    d = (t != t0)
                    N = t.shape[0]
                    C = t.shape[-1]
                    per_clip = d.reshape(N, -1).sum(1).tolist()
                    rows = d.reshape(-1, C).any(1).nonzero().flatten()
                    chans = d.reshape(-1, C).any(0).nonzero().flatten()
                    tf, t0f = t.float(), t0.float()
                    where = dict(block=name, shape=list(t.shape), n_diff=int(d.sum()), per_clip=per_clip,
                                 row_min=int(rows.min()), row_max=int(rows.max()), n_rows=int(rows.numel()),
                                 ch_min=int(chans.min()), ch_max=int(chans.max()), n_ch=int(chans.numel()),
                                 rows_head=rows[:24].tolist(), max_abs=float((tf - t0f).abs().max()))
                    break
            if clips or where:
                report["bad"].append(dict(rep=r, clips=clips, first_block=where))
    report["n_bad"] = len(report["bad"])
    line = json.dumps(report)
    print(line)
    if args.out:
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        with open(args.out, "w") as f:
            f.write(line + "\n")


if __name__ == "__main__":
    main()
