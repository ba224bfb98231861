import os, sys
import torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from trace_amd import ops, _lib as L
dev = "cuda:0"
lib = L.lib()
def bench(fn, n=20, reps=5):
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e30
    for _ in range(reps):
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record(); g.replay(); e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / n * 1e3)
    return best
torch.manual_seed(0)
SH = [(6272, 512, 1024, 0), (6272, 512, 2048, 0), (6272, 256, 1024, 0), (25088, 128, 512, 0), (6272, 1536, 1024, 0), (148960, 256, 512, 0),
      (6272, 768, 512, 0), (3136, 784, 256, 0), (1568, 512, 2048, 0)]
for M, N, K, res in SH:
    A = torch.randn((M, K), device=dev).bfloat16()
    B = (torch.randn((N, K), device=dev) * 0.05).bfloat16()
    bias = torch.randn((N,), device=dev)
    run = lambda: ops.gemm_nt(A, B, bias=bias, relu=True)
    out = []
    for rep in range(2):
        lib.trace_igemm_tune(0, 0, 0)
        out.append("def %.1f" % bench(run))
        for tile in (1, 2, 3, 4):
            for nst in (2, 3):
                lib.trace_igemm_tune(tile, nst, 1)
                out.append("%d/%d %.1f" % (tile, nst, bench(run)))
        out.append("|")
    lib.trace_igemm_tune(0, 0, 0)
    print("M=%6d N=%5d K=%5d  %s" % (M, N, K, "  ".join(out)), flush=True)
    del A, B
