"""Is the pipelined bench step bound by the GPU or by the host?  Times the bench workload in the fast mode
(a) as bench.py runs it, (b) with the step's host preparation (roi lists, cluster trees, pair lists: numpy) computed
once and reused - a DIAGNOSTIC that skips per-step work and is therefore not a benchmark figure - and (c) the host
preparation alone.  (b) is the GPU-side floor of the step; (a) - (b) is what the host costs."""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from trace_amd.model_builder_rel import Generalized_RCNN
from trace_amd.synthetic import make_inputs, randomize_model


def main(steps=100):
    dev = torch.device("cuda", 0)
    torch.manual_seed(1)
    model = Generalized_RCNN()
    randomize_model(model, 1)
    model.eval().to(dev)
    model.graph_step, model.side_stream, model.overlap_steps, model.split_stage = True, True, True, (2, 3)
    calib = torch.randn((1, 3, 16, 112, 112), generator=torch.Generator().manual_seed(1)) * 50
    model.Prd_RCNN.calibrate_bn(calib.to(dev))
    frames_c, blob_c, feat_c, rois = make_inputs(8, 32, 20, 224, seed=1)
    frames = frames_c.to(dev)
    blob = blob_c.to(dev).permute(0, 2, 3, 1).contiguous().to(torch.bfloat16)
    feat = feat_c.to(dev).permute(0, 2, 3, 1).contiguous().to(torch.bfloat16)

    def run(n):
        for _ in range(n):
            f, b, m_ = model.step_buffers()
            model.forward_clips(f, b, m_, rois, topk=100, prd_k=20)
        model.flush_steps()
        torch.cuda.synchronize()

    for _ in range(6):
        model.forward_clips(frames, blob, feat, rois, topk=100, prd_k=20)
    model.flush_steps()
    torch.cuda.synchronize()
    run(10)
    t0 = time.perf_counter()
    run(steps)
    a = (time.perf_counter() - t0) / steps
    orig = model._host_prep
    t0 = time.perf_counter()
    for _ in range(steps):
        prep = orig(rois, model.Prd_RCNN.out_frames(32), 1.0, 1.0)
    c = (time.perf_counter() - t0) / steps
    model._host_prep = lambda *a_, **k_: prep
    run(10)
    t0 = time.perf_counter()
    run(steps)
    b = (time.perf_counter() - t0) / steps
    model._host_prep = orig
    print("step as benched          %.3f ms" % (a * 1e3))
    print("host prep reused (diag)  %.3f ms   <- GPU-side floor" % (b * 1e3))
    print("host prep alone          %.3f ms" % (c * 1e3))


if __name__ == "__main__":
    main()
