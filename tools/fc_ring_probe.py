"""Weight-streaming FCs (fc1: 160 x 3072 x 100352, fc_r: 168 x 1024 x 100352, pre-tiled weights): time per call for
LDS ring depths / split-K factors (trace_igemm_tune), replayed from a captured graph with a cache-flushing write
between calls so the weights come from HBM as in the step."""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from trace_amd import ops
from trace_amd import _lib as L

dev = torch.device("cuda:0")


def timeit(fn, flush, iters=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        flush.fill_(1.0)            # 1 GiB write: evicts the weights from L2 / MALL
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    flush = torch.empty((256 << 20,), dtype=torch.float32, device=dev)
    for name, M, N, K, tile in (("fc1", 160, 3072, 100352, 12), ("fc_r", 168, 1024, 100352, 8)):
        a = (torch.randn((M, K), device=dev) * 0.1).to(torch.bfloat16)
        w = ops.tile_weights((torch.randn((N, K), device=dev) * 0.01).to(torch.bfloat16))
        bias = torch.zeros((N,), device=dev)
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        fn = lambda: ops.gemm_nt(a, w, bias=bias, relu=True, out=out, b_tiled=True)
        wb = 2.0 * N * K
        t = timeit(fn, flush)
        print("%-5s default            %7.1f us  %5.2f TB/s of weights" % (name, t, wb / t * 1e-6), flush=True)
        for nst in (3, 4):
            for ks in (0, 16, 21, 24, 32):
                L.lib().trace_igemm_tune(tile, nst, ks)
                try:
                    t = timeit(fn, flush)
                finally:
                    L.lib().trace_igemm_tune(0, 0, 0)
                print("%-5s ring %d split-K %2d  %7.1f us  %5.2f TB/s" % (name, nst, ks, t, wb / t * 1e-6), flush=True)
        del a, w


if __name__ == "__main__":
    main()
