"""GPU parity of the bf16 MFMA implicit-GEMM conv3d / GEMM kernels against a plain
PyTorch fp32 reference of the same op (run on CPU, on the bf16-rounded operands).
Tolerance: rtol 2e-2 of the output scale (north_star, bf16 vs fp32)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _pack_w(w):
    # (Cout,Cin,kt,kh,kw) -> (Cout, kt*kh*kw*Cin) tap-major, channel-minor
    return w.permute(0, 2, 3, 4, 1).reshape(w.shape[0], -1).contiguous()


def _close(got, ref, tol=2e-2):
    scale = float(ref.abs().max()) + 1e-6
    err = float((got - ref).abs().max())
    assert err <= tol * scale, "max err %g vs scale %g" % (err, scale)


CASES = [
    # N, Cin, T, H, W, Cout, k, stride, pad, dil
    (2, 64, 4, 14, 14, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1)),
    (2, 64, 4, 14, 14, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),
    (1, 64, 2, 9, 11, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)),
    (2, 128, 3, 16, 16, 128, (1, 3, 3), (1, 2, 2), (0, 1, 1), (1, 1, 1)),
    (1, 256, 2, 16, 16, 512, (1, 1, 1), (1, 2, 2), (0, 0, 0), (1, 1, 1)),
    (1, 128, 2, 14, 14, 128, (1, 3, 3), (1, 1, 1), (0, 2, 2), (1, 2, 2)),
    (1, 192, 3, 7, 7, 72, (3, 3, 3), (2, 1, 1), (1, 1, 1), (1, 1, 1)),   # odd sizes, K tail
    (3, 8, 2, 5, 5, 8, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)),       # tiny
    # 1x3x3 stride-1 convs big enough for the shared-patch kernel (sconv9): 14-row / 8-row tiles, dilation 1 / 2,
    # one / even / odd numbers of 64-channel chunks, ragged map edges and a ragged last cout tile
    (2, 128, 4, 40, 40, 192, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),
    (2, 192, 4, 30, 20, 320, (1, 3, 3), (1, 1, 1), (0, 2, 2), (1, 2, 2)),
    (4, 256, 4, 14, 14, 256, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),
    (4, 64, 4, 14, 14, 248, (1, 3, 3), (1, 1, 1), (0, 2, 2), (1, 2, 2)),
    (2, 64, 8, 56, 56, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),
]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("fuse", [False, True])
def test_conv3d_igemm(case, fuse):
    from trace_amd import ops
    N, Cin, T, H, W, Cout, k, s, p, d = case
    g = torch.Generator().manual_seed(1)
    x = torch.randn((N, Cin, T, H, W), generator=g).bfloat16()
    w = (torch.randn((Cout, Cin) + k, generator=g) / (Cin * k[0] * k[1] * k[2]) ** 0.5).bfloat16()
    b = torch.randn((Cout,), generator=g)
    ref = F.conv3d(x.float(), w.float(), b, stride=s, padding=p, dilation=d)
    res = None
    if fuse:
        res = torch.randn(ref.shape, generator=g).bfloat16()
        ref = torch.relu(ref + res.float())
    out = ops.conv3d(x.permute(0, 2, 3, 4, 1).contiguous().to(DEV), _pack_w(w).to(DEV), b.to(DEV), k, s, p, d,
                     residual=None if res is None else res.permute(0, 2, 3, 4, 1).contiguous().to(DEV), relu=fuse)
    got = out.float().permute(0, 4, 1, 2, 3).cpu()
    assert got.shape == ref.shape
    _close(got, ref)


@pytest.mark.parametrize("shape", [(2, 64, 4, 14, 14, 256), (1, 128, 2, 9, 11, 72), (3, 64, 8, 20, 20, 128)])
@pytest.mark.parametrize("res", [False, True])
def test_conv3d_pooled_epilogue_equals_conv_then_pool(shape, res):
    """TRACE_EPI_POOL_T2: MaxPool3d((2,1,1)) inside the 1x1x1 conv's epilogue == conv, then the pool kernel."""
    from trace_amd import ops
    N, Cin, T, H, W, Cout = shape
    g = torch.Generator().manual_seed(5)
    x = torch.randn((N, T, H, W, Cin), generator=g).bfloat16().to(DEV)
    w = (torch.randn((Cout, Cin), generator=g) / Cin ** 0.5).bfloat16().to(DEV)
    b = torch.randn((Cout,), generator=g).to(DEV)
    r = torch.randn((N, T, H, W, Cout), generator=g).bfloat16().to(DEV) if res else None
    want = ops.maxpool3d(ops.conv3d(x, w, b, (1, 1, 1), residual=r, relu=True), (2, 1, 1), (2, 1, 1))
    got = ops.conv3d(x, w, b, (1, 1, 1), residual=r, relu=True, pool_t2=True)
    assert got.shape == want.shape == (N, T // 2, H, W, Cout)
    assert torch.equal(got, want)
    # not eligible (3 frames): falls back to the two launches
    x3 = x[:, :1].repeat(1, 3, 1, 1, 1).contiguous()
    assert ops.conv3d(x3, w, b, (1, 1, 1), relu=True, pool_t2=True).shape[1] == 1


@pytest.mark.parametrize("shape", [(1000, 256, 64, 64), (6272, 384, 128, 192), (130, 72, 64, 128)])
def test_gemm_nt_two_sources(shape):
    """[A | A2] @ B^T (trace_gemm_bf16_nt_dual) against the fp32 product of the concatenated operands."""
    from trace_amd import ops
    M, N, K1, K2 = shape
    g = torch.Generator().manual_seed(9)
    wide = torch.randn((M, K1 + 32), generator=g).bfloat16().to(DEV)       # row-strided view as first source
    a1 = wide[:, :K1]
    a2 = torch.randn((M, K2), generator=g).bfloat16().to(DEV)
    w = (torch.randn((N, K1 + K2), generator=g) / (K1 + K2) ** 0.5).bfloat16().to(DEV)
    b = torch.randn((N,), generator=g).to(DEV)
    got = ops.gemm_nt_dual(a1, a2, w, bias=b, relu=True).float().cpu()
    ref = torch.relu(torch.cat([a1.float(), a2.float()], 1).cpu() @ w.float().cpu().t() + b.cpu())
    _close(got, ref)
    same = ops.gemm_nt(torch.cat([a1, a2], 1).contiguous(), w, bias=b, relu=True)
    assert torch.equal(same.float().cpu(), got)


def test_conv3d_stem_matches_strided_conv():
    """Stem: 3->64, k(5,7,7), s(2,2,2), p(2,3,3) on the halo-padded clip, incl. the folded
    temporal stride 4 (= conv stride 2 followed by the pool's frame skipping)."""
    from trace_amd import ops
    g = torch.Generator().manual_seed(2)
    N, T, H, W = 2, 8, 32, 32
    clip = torch.randn((N, T, 3, H, W), generator=g) * 50
    w = torch.randn((64, 3, 5, 7, 7), generator=g) * 0.02
    b = torch.randn((64,), generator=g)
    xb = clip.bfloat16().float()
    wb = w.bfloat16().float()
    ref = torch.relu(F.conv3d(xb.permute(0, 2, 1, 3, 4), wb, b, stride=(2, 2, 2), padding=(2, 3, 3)))
    # pack: (Cout, kt, kh, kwp=8, Cp=4)
    wp = torch.zeros((64, 5, 7, 8, 4))
    wp[:, :, :, :7, :3] = w.permute(0, 2, 3, 4, 1)
    wp = wp.reshape(64, -1).bfloat16().to(DEV)
    xp = ops.clip_to_ndhwc(clip.to(DEV), True, 4, (2, 2, 3, 3, 3, 5))
    assert xp.shape == (N, T + 4, H + 6, W + 8, 4)
    for st in (2, 4):
        To = (T + 4 - 5) // st + 1
        out = ops.conv3d_stem(xp, wp, b.to(DEV), 5, 7, 8, (st, 2, 2), (To, 16, 16))
        got = out.float().permute(0, 4, 1, 2, 3).cpu()
        _close(got, ref[:, :, ::(st // 2)])
        # the same kernel fed from the f32 clip (both clip layouts): bit-identical to the padded-copy path
        out2 = ops.conv3d_stem_f32(clip.to(DEV), True, wp, b.to(DEV), st, (To, 16, 16))
        assert torch.equal(out2, out)
        out3 = ops.conv3d_stem_f32(clip.permute(0, 2, 1, 3, 4).contiguous().to(DEV), False, wp, b.to(DEV), st, (To, 16, 16))
        assert torch.equal(out3, out)


def test_conv3d_stem_from_uint8_frames_is_bit_identical_to_the_f32_entry():
    """trace_conv3d_stem16_u8: decoded uint8 frames (N,T,H,W,C), mean subtracted on the fly (get_frames_blob,
    test_engine_rel.py:420-520).  Same pixels through the f32 entry -> the same bf16 outputs, bit for bit: odd
    sizes that leave halo on every side, both temporal strides, 1- and 3-channel frames, no mean."""
    from trace_amd import ops
    g = torch.Generator().manual_seed(5)
    mean3 = [102.9801, 115.9465, 122.7717]     # cfg.PIXEL_MEANS
    for (N, T, H, W, C) in ((2, 8, 32, 32, 3), (1, 5, 64, 96, 3), (1, 4, 32, 64, 1)):
        frames = torch.randint(0, 256, (N, T, H, W, C), generator=g, dtype=torch.uint8)
        w = torch.randn((64, 3, 5, 7, 7), generator=g) * 0.02
        b = torch.randn((64,), generator=g)
        wp = torch.zeros((64, 5, 7, 8, 4))
        wp[:, :, :, :7, :3] = w.permute(0, 2, 3, 4, 1)
        wp = wp.reshape(64, -1).bfloat16().to(DEV)
        for mean in (mean3[:C], None):
            m = torch.tensor(mean if mean is not None else [0.0] * C, dtype=torch.float32)
            clip = (frames.float() - m).permute(0, 1, 4, 2, 3).contiguous()        # (N,T,C,H,W) f32, as the reference feeds it
            for st in (2, 4):
                To = (T + 4 - 5) // st + 1
                if To < 1:
                    continue
                Ho, Wo = H // 2, W // 2
                ref = ops.conv3d_stem_f32(clip.to(DEV), True, wp, b.to(DEV), st, (To, Ho, Wo))
                got = ops.conv3d_stem_u8(frames.to(DEV), mean, wp, b.to(DEV), st, (To, Ho, Wo))
                assert torch.equal(got, ref), (N, T, H, W, C, mean is None, st, int((got != ref).sum()))
    with pytest.raises(RuntimeError):
        ops.conv3d_stem_u8(torch.zeros((1, 8, 32, 32, 3), dtype=torch.float32, device=DEV), None, wp, None, 2, (4, 16, 16))


@pytest.mark.parametrize("shape", [(300, 136, 200), (3136, 256, 784), (37, 1000, 64), (1, 8, 8), (160, 1024, 1024),
                                   (96, 384, 16384), (20, 64, 4096),      # split-K path
                                   (160, 384, 32768), (50, 200, 32768),   # weight-streaming shapes
                                   (150, 256, 65536), (176, 128, 65536)]) # ... on the 128c x 160p / 128c x 192p tiles
def test_gemm_nt(shape):
    from trace_amd import ops
    M, N, K = shape
    g = torch.Generator().manual_seed(3)
    A = torch.randn((M, K), generator=g).bfloat16()
    B = (torch.randn((N, K), generator=g) / K ** 0.5).bfloat16()
    bias = torch.randn((N,), generator=g)
    ref = A.float() @ B.float().t() * 0.5 + bias
    out = ops.gemm_nt(A.to(DEV), B.to(DEV), bias=bias.to(DEV), alpha=0.5, out_f32=True)
    _close(out.cpu(), ref)
    out2 = ops.gemm_nt(A.to(DEV), B.to(DEV), bias=bias.to(DEV), alpha=0.5, relu=True)
    _close(out2.float().cpu(), torch.relu(ref))


@pytest.mark.parametrize("shape", [(160, 384, 32768), (50, 256, 32768), (192, 128, 65536), (300, 256, 8192),
                                   (150, 256, 65536), (129, 128, 8192)])
def test_gemm_nt_pretiled_weights_and_register_fed_kernel(shape):
    """Weight-streaming paths: TRACE_EPI_WTILED (ops.tile_weights layout) and the register-fed kernel (tile id 10)
    give the row-major result."""
    from trace_amd import ops, _lib as L
    M, N, K = shape
    g = torch.Generator().manual_seed(5)
    A = torch.randn((M, K), generator=g).bfloat16().to(DEV)
    B = (torch.randn((N, K), generator=g) / K ** 0.5).bfloat16().to(DEV)
    bias = torch.randn((N,), generator=g).to(DEV)
    ref = ops.gemm_nt(A, B, bias=bias, relu=True, out_f32=True)
    got = ops.gemm_nt(A, ops.tile_weights(B), bias=bias, relu=True, out_f32=True, b_tiled=True)
    _close(got.cpu(), ref.cpu())
    L.lib().trace_igemm_tune(10, 0, 0)
    try:
        got2 = ops.gemm_nt(A, B, bias=bias, relu=True, out_f32=True)
    finally:
        L.lib().trace_igemm_tune(0, 0, 0)
    _close(got2.cpu(), ref.cpu())
    ref32 = torch.relu(A.float().cpu() @ B.float().cpu().t() + bias.cpu())
    _close(got.cpu(), ref32)


def test_gemm_nt_batched_and_transpose():
    from trace_amd import ops
    g = torch.Generator().manual_seed(4)
    A = torch.randn((3, 196, 64), generator=g).bfloat16()
    B = torch.randn((3, 52, 64), generator=g).bfloat16()
    out = ops.gemm_nt(A.to(DEV), B.to(DEV), alpha=1.0 / 52, out_f32=True)
    ref = torch.bmm(A.float(), B.float().transpose(1, 2)) / 52
    _close(out.cpu(), ref)
    t = ops.transpose(B.to(DEV))
    assert torch.equal(t.cpu(), B.transpose(1, 2).contiguous())
    tp = ops.transpose(B.to(DEV), pad_to=8)                       # zero-padded rows (52 -> 56)
    assert tp.shape == (3, 64, 56) and torch.equal(tp[:, :, :52].cpu(), B.transpose(1, 2)) and float(tp[:, :, 52:].abs().max()) == 0


def test_maxpool3d():
    from trace_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn((2, 16, 6, 12, 12), generator=g).bfloat16()
    xl = x.permute(0, 2, 3, 4, 1).contiguous().to(DEV)
    for k, s, p in (((1, 3, 3), (2, 2, 2), (0, 1, 1)), ((2, 1, 1), (2, 1, 1), (0, 0, 0)), ((1, 2, 2), (1, 2, 2), (0, 0, 0)),
                    ((1, 3, 3), (1, 2, 2), (0, 1, 1))):
        ref = F.max_pool3d(x.float(), k, s, p)
        got = ops.maxpool3d(xl, k, s, p).float().permute(0, 4, 1, 2, 3).cpu()
        assert torch.equal(got, ref)


def test_split_k_plan_is_fixed_by_trace_gemm_workspace_bytes():
    """A workspace of exactly trace_gemm_workspace_bytes(...) gives the bits of an arbitrarily large one (the plan - and
    with it the f32 summation order - only changes BELOW that size), and a smaller one still gives a valid result."""
    from trace_amd import _lib as L
    from trace_amd import ops
    torch.manual_seed(5)
    M, N, K = 96, 256, 16384
    A = (torch.randn((M, K), device=DEV) * 0.1).to(torch.bfloat16)
    B = (torch.randn((N, K), device=DEV) * 0.1).to(torch.bfloat16)
    need = int(L.lib().trace_gemm_workspace_bytes(M, N, K, K, K, N, 1, 0))
    assert need > 0 and need % (M * N * 4) == 0

    def run(nbytes):
        ws = torch.empty((max(nbytes, 16),), dtype=torch.uint8, device=DEV)
        out = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
        L.check(L.lib().trace_gemm_bf16_nt(L.ptr(A), L.ptr(B), M, N, K, K, K, N, 1, 0, 0, 0, 1.0, None, None, 0, L.ptr(out),
                                           L.ptr(ws) if nbytes else None, nbytes, L.stream()), "gemm")
        torch.cuda.synchronize()
        return out
    big, exact, small, none = run(64 << 20), run(need), run(need // 2), run(0)
    assert torch.equal(big, exact)
    ref = A.float() @ B.float().t()
    for got in (exact, small, none):
        assert float((got.float() - ref).norm() / ref.norm()) < 1e-2
