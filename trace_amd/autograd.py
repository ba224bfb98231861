"""Backward of the conv / GEMM family through the same HIP kernels (`loss.backward()` of
tools/train_net_step_rel.py:625 reaches the I3D stages the config leaves trainable, lib/i3d/resnet3d.py:746-767, and
the Linear layers of the relation head).

A convolution's two gradients are convolutions / GEMMs again, so no new device code is needed:

  data gradient    dx = conv3d(zero-stuffed dy, W flipped and transposed, stride 1, padding d (k - 1) - p)
                   - the implicit-GEMM kernel of the forward pass on another operand;
  weight gradient  dW[co][tap][ci] = sum_pos dy[pos][co] x[pos s + tap d - p][ci]: per tap one NT GEMM of dy^T (Cout, M)
                   with the tap's shifted input view transposed (Cin, M), f32 result (K = M positions, split-K inside);
  bias gradient    column sums of dy.

Operands are bf16 as in the forward pass (dy is rounded to bf16), accumulation f32, weight gradients come back in f32
for the f32 master weights.  The strided views / zero stuffing / transposes around the kernels are torch indexing and
`ops.transpose` - plumbing, not the product's arithmetic.  BatchNorm is handled by the caller (ConvModule.run_train):
folded into the weights in eval mode, torch's batch_norm over the raw conv output in training mode.
"""
import torch
import torch.nn.functional as F

from . import ops


def _triple_out(i, k, s, p, d):
    return (i + 2 * p - d * (k - 1) - 1) // s + 1


def conv3d_dgrad(gy, w_bf, x_shape, ksize, stride, padding, dilation):
    """gy (N,To,Ho,Wo,Cout) bf16, w_bf (Cout, taps*Cin) bf16 tap-major -> dx (N,Ti,Hi,Wi,Cin) bf16."""
    N, Ti, Hi, Wi, Cin = x_shape
    kt, kh, kw = ksize
    Cout = w_bf.shape[0]
    pads = []
    for k, p, d in zip(ksize, padding, dilation):
        q = d * (k - 1) - p
        if q < 0:
            raise RuntimeError("conv3d backward: padding %d exceeds dilation*(k-1) = %d" % (p, d * (k - 1)))
        pads.append(q)
    # zero-stuff dy back onto the stride-1 grid and append the rows the strided forward never reached
    sizes = []
    for i, k, s, p, d, o in zip((Ti, Hi, Wi), ksize, stride, padding, dilation, gy.shape[1:4]):
        sizes.append(i + 2 * p - d * (k - 1))      # length of the stride-1 output grid
    if tuple(stride) != (1, 1, 1) or tuple(sizes) != tuple(gy.shape[1:4]):
        up = torch.zeros((N, sizes[0], sizes[1], sizes[2], Cout), dtype=gy.dtype, device=gy.device)
        up[:, ::stride[0], ::stride[1], ::stride[2]][:, :gy.shape[1], :gy.shape[2], :gy.shape[3]] = gy
        gy = up
    wt = w_bf.view(Cout, kt, kh, kw, Cin).flip(1, 2, 3).permute(4, 1, 2, 3, 0).reshape(Cin, kt * kh * kw * Cout).contiguous()
    gx = ops.conv3d(gy.contiguous(), wt, None, ksize, (1, 1, 1), tuple(pads), dilation)
    if tuple(gx.shape) != tuple(x_shape):
        raise RuntimeError("conv3d backward: data gradient shape %s != input shape %s" % (tuple(gx.shape), tuple(x_shape)))
    return gx


def conv3d_wgrad(gy, x, ksize, stride, padding, dilation):
    """gy (N,To,Ho,Wo,Cout) bf16, x (N,Ti,Hi,Wi,Cin) bf16 -> dW (Cout, taps*Cin) f32, tap-major like the weights."""
    N, To, Ho, Wo, Cout = gy.shape
    Cin = x.shape[-1]
    kt, kh, kw = ksize
    M = N * To * Ho * Wo
    gyT = ops.transpose(gy.reshape(M, Cout), pad_to=8)                                   # (Cout, Mp), zero tail
    if tuple(padding) != (0, 0, 0):
        x = F.pad(x, (0, 0, padding[2], padding[2], padding[1], padding[1], padding[0], padding[0]))
    taps = []
    for a in range(kt):
        for b in range(kh):
            for c in range(kw):
                t0, h0, w0 = a * dilation[0], b * dilation[1], c * dilation[2]
                xs = x[:, t0:t0 + (To - 1) * stride[0] + 1:stride[0], h0:h0 + (Ho - 1) * stride[1] + 1:stride[1],
                       w0:w0 + (Wo - 1) * stride[2] + 1:stride[2], :]
                xsT = ops.transpose(xs.reshape(M, Cin), pad_to=8)                        # (Cin, Mp)
                taps.append(ops.gemm_nt(gyT, xsT, out_f32=True))                         # (Cout, Cin)
    return torch.stack(taps, 1).reshape(Cout, kt * kh * kw * Cin)


class _Conv3d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, ksize, stride, padding, dilation, relu):
        wb = w.detach().to(torch.bfloat16).contiguous()
        y = ops.conv3d(x.detach(), wb, None if bias is None else bias.detach(), ksize, stride, padding, dilation, relu=relu)
        ctx.save_for_backward(x.detach(), wb, y if relu else None)
        ctx.cfg = (tuple(ksize), tuple(stride), tuple(padding), tuple(dilation), relu, w.dtype)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, wb, y = ctx.saved_tensors
        ksize, stride, padding, dilation, relu, wdt = ctx.cfg
        gy = gy.to(torch.bfloat16)
        if relu:
            gy = gy * (y > 0)
        gy = gy.contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = conv3d_dgrad(gy, wb, x.shape, ksize, stride, padding, dilation)
        if ctx.needs_input_grad[1]:
            gw = conv3d_wgrad(gy, x, ksize, stride, padding, dilation).to(wdt)
        if ctx.needs_input_grad[2]:
            gb = gy.float().sum((0, 1, 2, 3))
        return gx, gw, gb, None, None, None, None, None


def conv3d(x, w, bias, ksize, stride=(1, 1, 1), padding=(0, 0, 0), dilation=(1, 1, 1), relu=False):
    """Differentiable ops.conv3d: x (N,T,H,W,Cin) bf16, w (Cout, taps*Cin) f32 (or bf16) master weights, tap-major,
    bias (Cout) f32 or None."""
    return _Conv3d.apply(x, w, bias, tuple(ksize), tuple(stride), tuple(padding), tuple(dilation), bool(relu))


class _GemmNT(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, relu):
        wb = w.detach().to(torch.bfloat16).contiguous()
        xb = x.detach().to(torch.bfloat16).contiguous()
        y = ops.gemm_nt(xb, wb, bias=None if bias is None else bias.detach(), relu=relu)
        ctx.save_for_backward(xb, wb, y if relu else None)
        ctx.cfg = (relu, w.dtype, x.dtype)
        return y

    @staticmethod
    def backward(ctx, gy):
        xb, wb, y = ctx.saved_tensors
        relu, wdt, xdt = ctx.cfg
        gy = gy.to(torch.bfloat16)
        if relu:
            gy = gy * (y > 0)
        gy = gy.contiguous()
        M, N = gy.shape
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            wT = ops.transpose(wb, pad_to=8)                                     # (K, Np)
            g = gy if wT.shape[1] == N else F.pad(gy, (0, wT.shape[1] - N))
            gx = ops.gemm_nt(g.contiguous(), wT).to(xdt)                         # (M, K)
        if ctx.needs_input_grad[1]:
            gw = ops.gemm_nt(ops.transpose(gy, pad_to=8), ops.transpose(xb, pad_to=8), out_f32=True).to(wdt)   # (N, K)
        if ctx.needs_input_grad[2]:
            gb = gy.float().sum(0)
        return gx, gw, gb, None


def gemm_nt(x, w, bias=None, relu=False):
    """Differentiable act(x @ w^T + bias): x (M,K) bf16, w (N,K) f32 (or bf16) master weights, K % 8 == 0."""
    return _GemmNT.apply(x, w, bias, bool(relu))
