"""Yardstick for the dense conv-as-GEMM loop: trace_amd's ops.gemm_nt against the vendor GEMM (torch.matmul ->
hipBLASLt) on the backbone's 1x1x1 conv shapes (M positions, N couts, K cins), bf16 in / bf16 out, bias + ReLU in
ours.  Prints µs per call and TFLOP/s for both; under rocprofv3 --kernel-trace the library's kernel names (tile
sizes) appear in the trace.  Measurement tool only - the product never calls the library GEMM."""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from trace_amd import ops

SHAPES = [  # (M, N, K): 8 clips x 16 frames, 224 px
    (200704, 64, 64), (200704, 256, 64), (200704, 64, 256),          # layer1 (56x56x8 frames... per clip batch)
    (50176, 128, 256), (50176, 512, 128), (50176, 128, 512), (50176, 512, 256),      # layer2
    (25088, 256, 512), (25088, 1024, 256), (25088, 256, 1024), (25088, 1024, 512), (25088, 512, 1024),  # layer3
    (6272, 512, 1024), (6272, 2048, 512), (6272, 512, 2048), (6272, 2048, 1024),     # layer4
    (6272, 1536, 1024), (25088, 768, 512),
]


def timeit(fn, iters=20, reps=5):
    """µs per call with the launches replayed from a captured graph (no host launch overhead in the figure)."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with ops.capture(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (iters * reps)


BIG = "--big-tiles" in sys.argv      # also time the 128c x 256p (5), 256c x 128p (6), 256 x 256 (7) tiles of the same loop


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    print("%-22s %10s %10s %8s %8s %6s" % ("M x N x K", "ours us", "lib us", "ours TF", "lib TF", "ratio"))
    only = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 and sys.argv[1][0].isdigit() else range(len(SHAPES))
    for M, N, K in [SHAPES[i] for i in only]:
        a = (torch.randn((M, K), generator=g) * 0.5).to(torch.bfloat16).to(dev)
        w = (torch.randn((N, K), generator=g) * 0.05).to(torch.bfloat16).to(dev)
        bias = torch.randn((N,), generator=g).to(dev)
        wt = w.t().contiguous()
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        t_ours = timeit(lambda: ops.gemm_nt(a, w, bias=bias, relu=True, out=out))
        big = []
        if BIG:
            from trace_amd import _lib as L
            for tile, nst in ((5, 2), (5, 3), (6, 2), (6, 3), (7, 2)):
                L.lib().trace_igemm_tune(tile, nst, 0)
                try:
                    big.append("%d/%d:%.1f" % (tile, nst, timeit(lambda: ops.gemm_nt(a, w, bias=bias, relu=True, out=out))))
                finally:
                    L.lib().trace_igemm_tune(0, 0, 0)
        t_nt = timeit(lambda: torch.matmul(a, w.t(), out=out))
        t_nn = timeit(lambda: torch.matmul(a, wt, out=out))
        t_lib = min(t_nt, t_nn)
        fl = 2.0 * M * N * K
        print("%-22s %10.1f %10.1f %8.0f %8.0f %6.2f" % ("%d x %d x %d" % (M, N, K), t_ours, t_lib, fl / t_ours * 1e-6,
                                                         fl / t_lib * 1e-6, t_ours / t_lib), " ".join(big), flush=True)


if __name__ == "__main__":
    main()
