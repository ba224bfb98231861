#!/usr/bin/env python3
"""Diagnostic: which host-side calls of one eager step become device copy nodes (hipMemcpyAsync -> __amd_rocclr_copyBuffer
in a rocprof trace)?  Wraps Tensor.copy_ / clone / contiguous / to and prints the call sites with byte counts.

    python tools/count_copies.py
"""
import collections
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from trace_amd.model_builder_rel import Generalized_RCNN
    from trace_amd.synthetic import make_inputs, randomize_model
    dev = torch.device("cuda", 0)
    m = Generalized_RCNN(conv_bodies=False)
    randomize_model(m, 1)
    m.eval().to(dev)
    frames, blob, feat, rois = make_inputs(8, 32, 20, 224, seed=1)
    frames, blob, feat = frames.to(dev), blob.to(dev), feat.to(dev)
    for _ in range(2):
        m.forward_clips(frames, blob, feat, rois, topk=100, prd_k=1)
    torch.cuda.synchronize()
    sites = collections.Counter()
    nbytes = collections.Counter()

    def wrap(name):
        orig = getattr(torch.Tensor, name)

        def f(self, *a, **k):
            out = orig(self, *a, **k)
            moved = name == "copy_" or (torch.is_tensor(out) and out.data_ptr() != self.data_ptr())
            if moved and self.is_cuda:
                fr = [s for s in traceback.extract_stack()[:-1] if "trace_amd" in s.filename][-1:]
                key = (name, "%s:%d" % (os.path.basename(fr[0].filename), fr[0].lineno) if fr else "?")
                sites[key] += 1
                nbytes[key] += (out if torch.is_tensor(out) else self).numel() * self.element_size()
            return out
        setattr(torch.Tensor, name, f)
        return orig
    origs = {n: wrap(n) for n in ("copy_", "clone", "contiguous", "to", "float", "bfloat16")}
    m.forward_clips(frames, blob, feat, rois, topk=100, prd_k=1)
    torch.cuda.synchronize()
    for n, o in origs.items():
        setattr(torch.Tensor, n, o)
    for key, c in sorted(sites.items(), key=lambda kv: -kv[1]):
        print("%4d x %-12s %-34s %10d B" % (c, key[0], key[1], nbytes[key]))
    print("total", sum(sites.values()))


if __name__ == "__main__":
    main()
