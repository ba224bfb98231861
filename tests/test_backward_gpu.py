"""Backward of the conv / GEMM family (trace_amd/autograd.py) against torch autograd in f32 on the CPU: the data,
weight and bias gradients of every conv shape the I3D bottlenecks use (1x1x1, 3x1x1, 1x3x3 with stride 2 and with
dilation 2, the strided 1x1x1 downsample), of a Linear, and of a whole Bottleneck3d (frozen BatchNorm statistics).
bf16 operands, f32 accumulation: relative L2 error < 2e-2 (north_star's bf16-vs-fp32 tolerance)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 2e-2


def rel(got, ref):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    return float((got - ref).norm() / ref.norm().clamp_min(1e-12))


@pytest.mark.parametrize("k,s,p,d,Cin,Cout,thw", [
    ((1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1), 64, 256, (4, 14, 14)),
    ((3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1), 128, 64, (4, 10, 12)),
    ((1, 3, 3), (1, 2, 2), (0, 1, 1), (1, 1, 1), 64, 64, (2, 15, 14)),      # odd size: one input row the stride never reaches
    ((1, 3, 3), (1, 1, 1), (0, 2, 2), (1, 2, 2), 64, 128, (2, 14, 14)),     # DC5 dilation
    ((1, 1, 1), (1, 2, 2), (0, 0, 0), (1, 1, 1), 64, 128, (2, 14, 14)),     # strided downsample
    ((1, 1, 1), (2, 1, 1), (0, 0, 0), (1, 1, 1), 64, 64, (5, 6, 6)),
])
@pytest.mark.parametrize("relu", [False, True])
def test_conv3d_backward_vs_torch(k, s, p, d, Cin, Cout, thw, relu):
    from trace_amd import autograd as tag
    g = torch.Generator().manual_seed(Cin + Cout + k[0] + 2 * k[1])
    N = 2
    x = torch.randn((N, Cin) + thw, generator=g)
    w = torch.randn((Cout, Cin) + k, generator=g) / (Cin * k[0] * k[1] * k[2]) ** 0.5
    b = torch.randn((Cout,), generator=g) * 0.1
    # reference on the bf16-rounded operands (what the kernels see), f32 arithmetic
    xr = x.bfloat16().float().requires_grad_(True)
    wr = w.bfloat16().float().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    y = F.conv3d(xr, wr, br, s, p, d)
    if relu:
        y = torch.relu(y)
    gy = torch.randn(y.shape, generator=g).bfloat16().float()
    y.backward(gy)
    xh = x.bfloat16().permute(0, 2, 3, 4, 1).contiguous().to(DEV).requires_grad_(True)
    wh = w.bfloat16().float().permute(0, 2, 3, 4, 1).reshape(Cout, -1).contiguous().to(DEV).requires_grad_(True)
    bh = b.to(DEV).requires_grad_(True)
    yh = tag.conv3d(xh, wh, bh, k, s, p, d, relu=relu)
    assert rel(yh.permute(0, 4, 1, 2, 3), y.detach()) < TOL
    # the ReLU mask is taken from each side's own forward result: compare where both agree on the sign
    yh.backward(gy.permute(0, 2, 3, 4, 1).contiguous().to(DEV).bfloat16())
    gw = wh.grad.view(Cout, k[0], k[1], k[2], Cin).permute(0, 4, 1, 2, 3)
    print("conv bwd k%s s%s d%s relu=%d: y %.4f dx %.4f dw %.4f db %.4f" % (k, s, d, relu, rel(yh.permute(0, 4, 1, 2, 3), y.detach()),
          rel(xh.grad.permute(0, 4, 1, 2, 3), xr.grad), rel(gw, wr.grad), rel(bh.grad, br.grad)))
    assert rel(xh.grad.permute(0, 4, 1, 2, 3), xr.grad) < TOL, "data gradient"
    assert wh.grad.dtype == torch.float32
    assert rel(gw, wr.grad) < TOL, "weight gradient"
    assert rel(bh.grad, br.grad) < TOL, "bias gradient"


@pytest.mark.parametrize("M,N,K", [(160, 1024, 256), (37, 40, 1024), (3040, 136, 256)])
def test_gemm_backward_vs_torch(M, N, K):
    from trace_amd import autograd as tag
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn((M, K), generator=g)
    w = torch.randn((N, K), generator=g) / K ** 0.5
    b = torch.randn((N,), generator=g) * 0.1
    xr = x.bfloat16().float().requires_grad_(True)
    wr = w.bfloat16().float().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    y = torch.relu(F.linear(xr, wr, br))
    gy = torch.randn(y.shape, generator=g).bfloat16().float()
    y.backward(gy)
    xh = x.bfloat16().to(DEV).requires_grad_(True)
    wh = w.bfloat16().float().to(DEV).requires_grad_(True)
    bh = b.to(DEV).requires_grad_(True)
    yh = tag.gemm_nt(xh, wh, bh, relu=True)
    assert rel(yh, y.detach()) < TOL
    yh.backward(gy.to(DEV).bfloat16())
    assert rel(xh.grad, xr.grad) < TOL and rel(wh.grad, wr.grad) < TOL and rel(bh.grad, br.grad) < TOL
    assert wh.grad.dtype == torch.float32 and xh.grad.dtype == torch.bfloat16


@pytest.mark.parametrize("stride,dilation,down", [(1, 1, False), (2, 1, True), (1, 2, True)])
def test_bottleneck_backward_vs_torch(stride, dilation, down):
    """One Bottleneck3d (lib/i3d/resnet3d.py:292-318), BatchNorm in eval mode: loss = sum(out * r); gradients of the
    input, of every conv weight and of the BN affine parameters against the module's own torch path on the CPU."""
    from trace_amd.resnet3d import Bottleneck3d, ConvModule
    torch.manual_seed(3 + stride + dilation)
    inpl, planes = 128, 64
    ds = ConvModule(inpl, planes * 4, 1, stride=(1, stride, stride), bias=False, norm_cfg=dict(type='BN3d'), act_cfg=None) if down else None
    if not down:
        inpl = planes * 4
    blk = Bottleneck3d(inpl, planes, spatial_stride=stride, dilation=dilation, downsample=ds)
    with torch.no_grad():
        for m in blk.modules():
            if isinstance(m, torch.nn.BatchNorm3d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.1)
                m.running_mean.normal_(0, 0.1)
                m.running_var.uniform_(0.5, 1.5)
    blk.eval()
    x = torch.randn((2, inpl, 4, 14, 14))
    xr = x.bfloat16().float().requires_grad_(True)

    # The reference follows the HIP path's forward numerics (folded weights and every stored activation rounded to bf16,
    # straight-through) with f32 arithmetic and f32 gradients: a ReLU mask that differs on 0.1 % of the elements -
    # what an unmatched 2^-9 rounding does - already moves a gradient by sqrt(0.001) = 3 % in L2, which would say
    # nothing about the backward kernels.
    def st_round(t):
        return t + (t.bfloat16().float() - t).detach()

    def ref_conv(m, t, relu):
        scale = m.bn.weight / torch.sqrt(m.bn.running_var + m.bn.eps)
        shift = m.bn.bias - m.bn.running_mean * scale
        wf = st_round(m.conv.weight * scale.view(-1, 1, 1, 1, 1))
        y = F.conv3d(t, wf, shift, m.conv.stride, m.conv.padding, m.conv.dilation)
        return st_round(torch.relu(y) if relu else y)

    identity = xr if blk.downsample is None else ref_conv(blk.downsample, xr, False)
    out = torch.relu(st_round(ref_conv(blk.conv3, ref_conv(blk.conv2, ref_conv(blk.conv1, xr, True), True), False) + identity))
    r = torch.randn(out.shape).bfloat16().float()
    (out * r).sum().backward()
    ref = {n: p.grad.clone() for n, p in blk.named_parameters()}
    gx_ref = xr.grad.clone()
    blk.zero_grad()
    blk.to(DEV)
    xh = x.bfloat16().permute(0, 2, 3, 4, 1).contiguous().to(DEV).requires_grad_(True)
    oh = blk.run_train(xh)
    assert rel(oh.permute(0, 4, 1, 2, 3), out.detach()) < TOL
    (oh.float() * r.permute(0, 2, 3, 4, 1).to(DEV)).sum().backward()
    errs = {n: rel(p.grad, ref[n]) for n, p in blk.named_parameters()}
    print("bottleneck bwd: dx %.4f | %s" % (rel(xh.grad.permute(0, 4, 1, 2, 3), gx_ref), " ".join("%s %.4f" % kv for kv in errs.items())))
    assert rel(xh.grad.permute(0, 4, 1, 2, 3), gx_ref) < TOL, "input gradient"
    for n, e in errs.items():
        assert e < TOL, (n, e)


def test_nonlocal_backward_vs_torch():
    """NonLocal3d (dot_product, sub_sample, BN on conv_out; mmcv 1.x semantics as oracle/i3d_ref.py restates them): the
    block has no ReLU, so the module's own f32 torch path is the reference as it stands."""
    from trace_amd.resnet3d import NonLocal3d
    torch.manual_seed(11)
    C = 128
    nl = NonLocal3d(C, sub_sample=True, use_scale=False, mode='dot_product', norm_cfg=dict(type='BN3d'))
    with torch.no_grad():
        bn = nl.conv_out.bn
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_(0, 0.1)
        bn.running_mean.normal_(0, 0.1)
        bn.running_var.uniform_(0.5, 1.5)
    nl.eval()
    x = torch.randn((2, C, 2, 6, 6))
    xr = x.bfloat16().float().requires_grad_(True)
    out = nl(xr)
    r = torch.randn(out.shape).bfloat16().float()
    (out * r).sum().backward()
    ref = {n: p.grad.clone() for n, p in nl.named_parameters()}
    gx_ref = xr.grad.clone()
    nl.zero_grad()
    nl.to(DEV)
    xh = x.bfloat16().permute(0, 2, 3, 4, 1).contiguous().to(DEV).requires_grad_(True)
    oh = nl.run_train(xh)
    assert rel(oh.permute(0, 4, 1, 2, 3), out.detach()) < TOL
    (oh.float() * r.permute(0, 2, 3, 4, 1).to(DEV)).sum().backward()
    errs = {n: rel(p.grad, ref[n]) for n, p in nl.named_parameters()}
    print("non-local bwd: dx %.4f | %s" % (rel(xh.grad.permute(0, 4, 1, 2, 3), gx_ref), " ".join("%s %.4f" % kv for kv in errs.items())))
    assert rel(xh.grad.permute(0, 4, 1, 2, 3), gx_ref) < TOL
    for n, e in errs.items():
        # phi / g pass a (1,2,2) max-pool: where the bf16 path and the f32 reference pick different window elements
        # (near-ties) a whole gradient entry moves, as with ReLU masks (measured: phi 2.6e-2, g 7e-3, the rest <= 5e-3)
        assert e < (4e-2 if n.startswith(("phi.", "g.")) else TOL), (n, e)


@pytest.mark.parametrize("ksize,stride,cin,cout", [((1, 1, 1), (1, 1, 1), 64, 256), ((3, 1, 1), (1, 1, 1), 128, 64),
                                                   ((1, 3, 3), (1, 2, 2), 64, 64)])
def test_convmodule_batch_statistics_vs_torch(ksize, stride, cin, cout):
    """ConvModule.run_train with its BatchNorm3d in TRAINING mode (the reference's I3D config: norm_eval=False,
    lib/i3d/resnet3d.py:804-811): output, running statistics and every gradient against nn.Conv3d + nn.BatchNorm3d +
    ReLU in f32 on the CPU (torch reference of a floating-point path; the conv output is rounded to bf16 before the
    normalisation on the HIP side, straight-through in the reference as in the bottleneck test)."""
    from trace_amd.resnet3d import ConvModule
    torch.manual_seed(11 + cin)
    pad = tuple((k - 1) // 2 for k in ksize)
    m = ConvModule(cin, cout, ksize, stride=stride, padding=pad, bias=False, norm_cfg=dict(type='BN3d'), act_cfg=dict(type='ReLU'))
    with torch.no_grad():
        m.conv.weight.mul_(2.0)
        m.bn.weight.uniform_(0.5, 1.5)
        m.bn.bias.normal_(0, 0.3)
    m.train()
    x = torch.randn((2, cin, 4, 14, 14)) * 1.5 + 0.3
    xr = x.bfloat16().float().requires_grad_(True)
    wq = m.conv.weight + (m.conv.weight.bfloat16().float() - m.conv.weight).detach()
    y = F.conv3d(xr, wq, None, m.conv.stride, m.conv.padding, m.conv.dilation)
    y = y + (y.bfloat16().float() - y).detach()
    rm, rv = m.bn.running_mean.clone(), m.bn.running_var.clone()
    out = torch.relu(F.batch_norm(y, rm, rv, m.bn.weight, m.bn.bias, True, m.bn.momentum, m.bn.eps))
    r = torch.randn(out.shape).bfloat16().float()
    (out * r).sum().backward()
    ref = {n: p.grad.clone() for n, p in m.named_parameters()}
    gx_ref = xr.grad.clone()
    m.zero_grad()
    m.to(DEV)
    xh = x.bfloat16().permute(0, 2, 3, 4, 1).contiguous().to(DEV).requires_grad_(True)
    oh = m.run_train(xh)
    assert rel(oh.permute(0, 4, 1, 2, 3), out.detach()) < TOL
    assert rel(m.bn.running_mean, rm) < TOL and rel(m.bn.running_var, rv) < TOL and int(m.bn.num_batches_tracked) == 1
    (oh.float() * r.permute(0, 2, 3, 4, 1).to(DEV)).sum().backward()
    assert rel(xh.grad.permute(0, 4, 1, 2, 3), gx_ref) < TOL, "input gradient"
    for n, p in m.named_parameters():
        assert rel(p.grad, ref[n]) < TOL, (n, rel(p.grad, ref[n]))
    # eval mode afterwards: the fold uses the statistics the training step left behind
    m.eval()
    with torch.no_grad():
        ev = m.run_train(xh.detach())
        want = torch.relu(F.batch_norm(y.detach(), rm, rv, m.bn.weight.cpu(), m.bn.bias.cpu(), False, 0.0, m.bn.eps))
    assert rel(ev.permute(0, 4, 1, 2, 3), want) < TOL


def test_resnet3d_train_mode_uses_batch_statistics():
    """model.train() with the reference's config (norm_eval=False, frozen_stages=0): the stem stays in eval mode, the
    BatchNorm3d of layers 1-4 normalise with batch statistics - running statistics move, the output differs from the
    eval-mode forward, gradients reach every parameter behind the stem."""
    from trace_amd.resnet3d import ResNet3d, i3d_dc5_config
    from trace_amd.synthetic import randomize_model
    m = ResNet3d(**i3d_dc5_config())
    randomize_model(m, 4)
    clip = torch.randn((2, 3, 8, 64, 64), generator=torch.Generator().manual_seed(10)) * 50
    m.calibrate_bn(clip)
    m.to(DEV)
    clip = clip.to(DEV)
    m.eval()
    with torch.no_grad():
        ev = m.forward_train(clip).float()
    m.train()
    assert not m.conv1.bn.training and m.layer1[0].conv1.bn.training and m.layer4[2].conv3.bn.training
    before = {n: b.clone() for n, b in m.named_buffers() if n.endswith("running_mean")}
    out = m.forward_train(clip)
    assert bool(torch.isfinite(out.float()).all()) and rel(out.detach().float(), ev) > 1e-3
    moved = [n for n, b in m.named_buffers() if n.endswith("running_mean") and not torch.equal(b, before[n])]
    assert moved and all(not n.startswith("conv1.") for n in moved) and any(n.startswith("layer4.") for n in moved)
    out.float().square().mean().backward()
    for n, p in m.named_parameters():
        if n.startswith("conv1."):
            assert p.grad is None, n
        else:
            assert p.grad is not None and bool(torch.isfinite(p.grad).all()), n
    m.eval()


def test_resnet3d_forward_train_reaches_the_unfrozen_stages():
    """ResNet3d.forward_train with frozen_stages = 2 (lib/modeling_rel/model_builder_rel.py:166-169): same output as the
    inference path (different launch fusion, so bf16-close, not bitwise), gradients on every parameter of layer3 / layer4
    and on none of the stem / layer1 / layer2."""
    from trace_amd.resnet3d import ResNet3d, i3d_dc5_config
    from trace_amd.synthetic import randomize_model
    cfg = i3d_dc5_config()
    cfg["frozen_stages"] = 2
    m = ResNet3d(**cfg)
    randomize_model(m, 2)
    clip = torch.randn((1, 3, 8, 64, 64), generator=torch.Generator().manual_seed(9)) * 50
    m.calibrate_bn(clip)        # O(1) activations: the non-local blocks are cubic in their input, random statistics overflow
    m.eval()
    m.to(DEV)
    clip = clip.to(DEV)
    want = m.forward_ndhwc(clip).float()
    out = m.forward_train(clip)
    assert out.requires_grad and tuple(out.shape) == tuple(want.shape)
    assert rel(out.detach(), want) < 3e-2
    out.float().square().mean().backward()
    for n, p in m.named_parameters():
        trainable = n.startswith("layer3.") or n.startswith("layer4.")
        if trainable:
            assert p.grad is not None and bool(torch.isfinite(p.grad).all()), n
        else:
            assert p.grad is None, n
    g = [p.grad.float().norm().item() for n, p in m.named_parameters() if n.startswith("layer3.0.conv1.conv")]
    assert g and g[0] > 0
