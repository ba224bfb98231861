"""Measured ceilings of this GPU beside the vendor figures.  What HBM rate do plain streaming kernels reach?  (yardstick for the weight-streaming FCs: fc1 reads
616 MB of weights per step).  Times stock torch kernels from a captured graph: a read-only reduction, a copy, a fill."""
import torch

dev = "cuda:0"


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / (3 * iters)


for mb in (616, 2048):
    n = mb * (1 << 20) // 2
    x = torch.ones((n,), dtype=torch.bfloat16, device=dev)
    y = torch.empty_like(x)
    xi = x.view(torch.int32)
    t = timeit(lambda: xi.sum())
    print("%5d MB  read-only sum   %7.1f us  %5.2f TB/s" % (mb, t * 1e6, 2 * n / t * 1e-12))
    t = timeit(lambda: torch.max(xi))
    print("%5d MB  read-only max   %7.1f us  %5.2f TB/s" % (mb, t * 1e6, 2 * n / t * 1e-12))
    t = timeit(lambda: y.copy_(x))
    print("%5d MB  copy            %7.1f us  %5.2f TB/s (read + write)" % (mb, t * 1e6, 4 * n / t * 1e-12))
    t = timeit(lambda: y.fill_(0))
    print("%5d MB  fill            %7.1f us  %5.2f TB/s" % (mb, t * 1e6, 2 * n / t * 1e-12))

# the matrix cores' measured ceiling on a shape that fills them (vendor GEMM, bf16 in, f32 accumulate)
for n in (4096, 8192):
    a = torch.randn((n, n), device=dev).to(torch.bfloat16)
    b = torch.randn((n, n), device=dev).to(torch.bfloat16)
    c = torch.empty((n, n), dtype=torch.bfloat16, device=dev)
    t = timeit(lambda: torch.matmul(a, b, out=c), iters=5)
    print("matmul %d^3 bf16     %7.1f us  %6.0f TFLOP/s" % (n, t * 1e6, 2.0 * n ** 3 / t * 1e-12))
