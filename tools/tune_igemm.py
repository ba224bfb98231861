#!/usr/bin/env python3
"""Sweep the igemm launch configurations (tile x LDS ring depth x split-K) over every conv / GEMM
shape of the bench workload (B clips x 32f x 224^2, 20 objects) on the GPU and print, per shape, the
default configuration's time next to the best one.  Feeds the shape table in csrc/igemm.hip.

    python tools/tune_igemm.py [B] [--quick]
"""
import itertools
import os
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from trace_amd import ops  # noqa: E402
from trace_amd import _lib as L  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 8
QUICK = "--quick" in sys.argv
dev = "cuda:0"


def bench(fn, n=20, reps=3):
    """GPU time per launch: n launches captured into a HIP graph and replayed (no host launch cost in
    the measurement, as in the product's captured step)."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with ops.capture(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(reps):
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record()
        g.replay()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / n * 1e3)
    return best  # us


CONVS = [
    ('l1 64->64 3x1x1', 64, 8, 56, 56, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1), 1),
    ('l1 256->64 3x1x1', 256, 8, 56, 56, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1), 2),
    ('l1 64->64 1x3x3', 64, 8, 56, 56, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1), 3),
    ('l1 64->256 1x1', 64, 8, 56, 56, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1), 4),
    ('l2 256->128 3x1x1', 256, 4, 56, 56, 128, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1), 1),
    ('l2 128->128 3x3 s2', 128, 4, 56, 56, 128, (1, 3, 3), (1, 2, 2), (0, 1, 1), (1, 1, 1), 1),
    ('l2 256->512 1x1 s2', 256, 4, 56, 56, 512, (1, 1, 1), (1, 2, 2), (0, 0, 0), (1, 1, 1), 1),
    ('l2 128->512 1x1', 128, 4, 28, 28, 512, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1), 4),
    ('l2 512->128 1x1', 512, 4, 28, 28, 128, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1), 2),
    ('l2 512->128 3x1x1', 512, 4, 28, 28, 128, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1), 1),
    ('l2 128->128 3x3', 128, 4, 28, 28, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1), 3),
    ('l2 NL 512->256 1x1', 512, 4, 28, 28, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1), 6),
    ('l2 NL 256->512 1x1', 256, 4, 28, 28, 512, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1), 2),
    ('l3 512->256 3x1x1', 512, 4, 28, 28, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1), 1),
    ('l3 256->256 3x3 s2', 256, 4, 28, 28, 256, (1, 3, 3), (1, 2, 2), (0, 1, 1), (1, 1, 1), 1),
    ('l3 512->1024 1x1 s2', 512, 4, 28, 28, 1024, (1, 1, 1), (1, 2, 2), (0, 0, 0), (1, 1, 1), 1),
    ('l3 256->1024 1x1', 256, 4, 14, 14, 1024, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1), 6),
    ('l3 1024->256 1x1', 1024, 4, 14, 14, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1), 3),
    ('l3 1024->256 3x1x1', 1024, 4, 14, 14, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1), 2),
    ('l3 256->256 3x3', 256, 4, 14, 14, 256, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1), 5),
    ('l3 NL 1024->512 1x1', 1024, 4, 14, 14, 512, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1), 10),
    ('l3 NL 512->1024 1x1', 512, 4, 14, 14, 1024, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1), 3),
    ('l4 512->512 3x3 d2', 512, 4, 14, 14, 512, (1, 3, 3), (1, 1, 1), (0, 2, 2), (1, 2, 2), 3),
    ('l4 512->2048 1x1', 512, 4, 14, 14, 2048, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1), 3),
    ('l4 1024->2048 1x1', 1024, 4, 14, 14, 2048, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1), 1),
    ('l4 2048->512 3x1x1', 2048, 4, 14, 14, 512, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1), 1),
    ('l4 2048->512 1x1', 2048, 4, 14, 14, 512, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1), 1),
]
# plain GEMMs (name, batch, M, N, K, count): non-local products and the relation-head layers
n_obj, P, Tn = 20 * B, 380 * B, 4
GEMMS = [
    ('NL2 qk', B, 3136, 784, 256, 2), ('NL2 pv', B, 3136, 256, 784, 2),
    ('NL3 qk', B, 784, 200, 512, 3), ('NL3 pv', B, 784, 512, 200, 3),
    ('fc1 box|s|o', 1, n_obj, 3072, 100352, 1), ('fc_r0', 1, n_obj, 1024, 100352, 1),
    ('fc2 1024', 1, n_obj, 1024, 1024, 4), ('tail fc2', 1, n_obj * Tn, 1024, 2048, 2),
    ('kmap/vmap', 1, B * 196, 256, 2048, 2), ('inferAB', 1, P * 49, 256, 512, 1),
    ('sp_fc', 1, P, 256, 12544, 1), ('rel 256->132', 1, P, 136, 256, 2),
]


def sweep(run, batch, name, M, N, K, cnt, flops):
    lib = L.lib()
    lib.trace_igemm_tune(0, 0, 0)
    t_def = bench(run)
    best = (t_def, (0, 0, 0))
    tiles = (1, 2, 3, 4, 5, 6, 7) if N > 128 else ((1, 2, 3, 4, 5) if N > 64 else (1, 2))
    ksp = (1, 2, 4, 8) if batch == 1 and K >= 512 else (1,)
    if M <= 1024 and K >= 8192:
        ksp = (0, 4, 8, 16)
    nsts = (2, 3) if QUICK else (2, 3, 4)
    for tile, nst, ks in itertools.product(tiles, nsts, ksp):
        lds = {1: 128, 2: 192, 3: 192, 4: 256, 5: 384, 6: 384, 7: 512}[tile] * 128 * nst
        if lds > 160 * 1024:
            continue
        lib.trace_igemm_tune(tile, nst, ks)
        t = bench(run)
        if t < best[0]:
            best = (t, (tile, nst, ks))
    lib.trace_igemm_tune(0, 0, 0)
    print('%-22s M=%7d N=%5d K=%6d x%2d  default %7.1f us %6.0f TF/s | best %7.1f us %6.0f TF/s  cfg tile=%d nst=%d ksplit=%d'
          % (name, M, N, K, cnt, t_def, flops / t_def / 1e6, best[0], flops / best[0] / 1e6, *best[1]), flush=True)
    return t_def * cnt, best[0] * cnt


tot_d = tot_b = 0.0
for name, cin, T, H, W, cout, k, s, p, d, cnt in CONVS:
    x = torch.randn((B, T, H, W, cin), device=dev).bfloat16()
    w = (torch.randn((cout, k[0] * k[1] * k[2] * cin), device=dev) * 0.05).bfloat16()
    b = torch.randn((cout,), device=dev)
    y = ops.conv3d(x, w, b, k, s, p, d, relu=True)
    K = k[0] * k[1] * k[2] * cin
    a, c = sweep(lambda: ops.conv3d(x, w, b, k, s, p, d, relu=True), 1, name, y.numel() // cout, cout, K, cnt,
                 2.0 * y.numel() * K)
    tot_d += a
    tot_b += c
print('convs: default %.1f us, best %.1f us' % (tot_d, tot_b))
gd = gb = 0.0
for name, batch, M, N, K, cnt in GEMMS:
    A = torch.randn((batch, M, K) if batch > 1 else (M, K), device=dev).bfloat16()
    Bm = (torch.randn((batch, N, K) if batch > 1 else (N, K), device=dev) * 0.05).bfloat16()
    a, c = sweep(lambda: ops.gemm_nt(A, Bm), batch, name, M, N, K, cnt, 2.0 * batch * M * N * K)
    gd += a
    gb += c
    del A, Bm
print('gemms: default %.1f us, best %.1f us' % (gd, gb))
