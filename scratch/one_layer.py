"""Run ONE conv / GEMM shape repeatedly (for rocprofv3 --pmc passes).
usage: one_layer.py <name> [tile nst ksplit] ; names: l4_3x3, l3nl, inferAB, l1c2, fc1"""
import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from trace_amd import ops, _lib as L
dev = "cuda:0"
name = sys.argv[1]
cfg = [int(a) for a in sys.argv[2:5]] if len(sys.argv) >= 5 else [0, 0, 0]
B = 8
torch.manual_seed(0)
def conv(cin, T, H, W, cout, k, s, p, d):
    x = torch.randn((B, T, H, W, cin), device=dev).bfloat16()
    w = (torch.randn((cout, k[0] * k[1] * k[2] * cin), device=dev) * 0.05).bfloat16()
    b = torch.randn((cout,), device=dev)
    return lambda: ops.conv3d(x, w, b, k, s, p, d, relu=True)
def gemm(M, N, K):
    A = torch.randn((M, K), device=dev).bfloat16()
    Bm = (torch.randn((N, K), device=dev) * 0.05).bfloat16()
    return lambda: ops.gemm_nt(A, Bm)
fn = {"l4_3x3": lambda: conv(512, 4, 14, 14, 512, (1, 3, 3), (1, 1, 1), (0, 2, 2), (1, 2, 2)),
      "l3nl": lambda: conv(1024, 4, 14, 14, 512, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)),
      "l1c2": lambda: conv(64, 8, 56, 56, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),
      "l1c1": lambda: conv(256, 8, 56, 56, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1)),
      "l1c3": lambda: conv(64, 8, 56, 56, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)),
      "inferAB": lambda: gemm(148960, 256, 512),
      "big": lambda: gemm(8192, 8192, 8192),
      "fc1": lambda: gemm(160, 3072, 100352)}[name]()
L.lib().trace_igemm_tune(*cfg)
for _ in range(int(os.environ.get("REPS", "20"))):
    fn()
torch.cuda.synchronize()
