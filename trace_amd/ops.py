"""Tensor-level wrappers of the channels-last bf16 kernels in libtrace_amd.so.

PyTorch is used for device memory and streams only; every op below is one call
through the C-ABI (include/trace_amd.h).  Tensors must live on the GPU.

Launch plans: the I3D backbone is ~130 launches of fixed shape per batch.  `record()` captures the
(C entry point, argument tuple) of every launch made inside it together with the tensors they touch;
`Plan.replay()` re-issues exactly those calls on the current stream with no Python/torch work per
launch (same role as a HIP graph, but launch-by-launch so the per-launch event trace still works).
"""
import ctypes

import os

import torch

from . import _lib as L

EPI_RELU = 1
EPI_RESIDUAL = 2
EPI_OUT_F32 = 4

# When set to a list, every igemm launch (conv3d / conv3d_stem / gemm_nt) is bracketed by a pair of
# events on the launch stream and (start, end, algorithmic_flops) is appended: bench.py's roofline leg.
IGEMM_TRACE = None
# diagnostics (tools/race_probe.py): a list collects (name, tensor) of the stem and every backbone block as they are launched
PROBE = None



class capture:
    """`with ops.capture(g):` = `with torch.cuda.graph(g):` with Python's cyclic garbage collector out of the way.
    A torch.cuda.CUDAGraph that dies INSIDE another graph's stream capture runs hipGraphExecDestroy + hipDeviceSynchronize
    there (its destructor, ROCm build), which the capture does not survive: an abort in the destructor, or a captured graph
    that later segfaults in hipGraphLaunch.  Graph objects in dead reference cycles die whenever the collector happens to
    run (torch 2.10 no longer collects at capture entry), so: collect now, on the idle device, and keep the collector off
    until the capture has ended."""

    def __init__(self, graph, **kw):
        self.graph, self.kw = graph, kw

    def __enter__(self):
        import gc
        torch.cuda.synchronize()
        gc.collect()
        self.was = gc.isenabled()
        gc.disable()
        self.ctx = torch.cuda.graph(self.graph, **self.kw)
        try:
            return self.ctx.__enter__()
        except BaseException:
            if self.was:
                gc.enable()
            raise

    def __exit__(self, *a):
        import gc
        try:
            return self.ctx.__exit__(*a)
        finally:
            if self.was:
                gc.enable()


_WS = {}
WORKSPACE_BYTES = 64 << 20
_RECORDING = None


_WS_SLOT = 0


class workspace_slot:
    """Launches made inside use split-K scratch `slot` of their device: two streams that run GEMMs
    concurrently (the head's side stream next to the backbone) must not share one scratch buffer."""

    def __init__(self, slot):
        self.slot = slot

    def __enter__(self):
        global _WS_SLOT
        self.prev, _WS_SLOT = _WS_SLOT, self.slot
        return self

    def __exit__(self, *a):
        global _WS_SLOT
        _WS_SLOT = self.prev
        return False


_WS_BASE = 0


class workspace_base:
    """Launches made inside use scratch slots base + {0, 1}: step graphs that run concurrently on different
    streams (model pipeline depth 2) must not share split-K scratch."""

    def __init__(self, base):
        self.base = base

    def __enter__(self):
        global _WS_BASE
        self.prev, _WS_BASE = _WS_BASE, self.base
        return self

    def __exit__(self, *a):
        global _WS_BASE
        _WS_BASE = self.prev
        return False


def workspace(device):
    """Per-device (and per slot) split-K scratch handed to the igemm entry points (the library allocates
    nothing).  Calls that share a buffer must be ordered on one stream."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), _WS_BASE + _WS_SLOT)
    ws = _WS.get(key)
    if ws is None:
        ws = torch.empty((WORKSPACE_BYTES,), dtype=torch.uint8, device=torch.device("cuda", key[0]))
        _WS[key] = ws
    return ws


def _raw_stream(index):
    return torch._C._cuda_getCurrentRawStream(index)


def _launch(name, fn, args, dev, flops=None):
    """One C-ABI launch on the current stream of `dev` (args exclude the trailing stream)."""
    idx = dev.index
    if idx is None:
        idx = torch.cuda.current_device()
    if idx != torch.cuda.current_device():
        with torch.cuda.device(idx):
            return _launch(name, fn, args, torch.device("cuda", idx), flops)
    stream = _raw_stream(idx)
    if IGEMM_TRACE is not None and flops is not None:
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record()
        rc = fn(*args, stream)
        e.record()
        IGEMM_TRACE.append((s, e, flops, 0))
    else:
        rc = fn(*args, stream)
    if rc != 0:
        L.check(rc, name)
    if _RECORDING is not None:
        _RECORDING.calls.append((fn, args, name, flops))


class Plan:
    """A recorded launch sequence (see module docstring)."""

    def __init__(self):
        self.calls = []
        self.keep = []
        self.out = None
        self.device_index = None

    def replay(self):
        stream = _raw_stream(self.device_index)
        trace = IGEMM_TRACE
        for fn, args, name, flops in self.calls:
            if trace is not None and flops is not None:
                s = torch.cuda.Event(enable_timing=True)
                e = torch.cuda.Event(enable_timing=True)
                s.record()
                rc = fn(*args, stream)
                e.record()
                trace.append((s, e, flops, 0))
            else:
                rc = fn(*args, stream)
            if rc != 0:
                L.check(rc, name)
        return self.out


class record:
    """Context manager: `with record(device) as plan: out = f(...)`; tensors created by torch inside
    stay alive as long as the plan does (torch's caching allocator cannot hand them out again)."""

    def __init__(self, device, keep=()):
        self.plan = Plan()
        self.plan.device_index = device.index if device.index is not None else torch.cuda.current_device()
        self.plan.keep.extend(keep)

    def __enter__(self):
        global _RECORDING
        if _RECORDING is not None:
            raise RuntimeError("nested ops.record()")
        _RECORDING = self.plan
        return self.plan

    def __exit__(self, *a):
        global _RECORDING
        _RECORDING = None
        return False


def _keep(*tensors):
    if _RECORDING is not None:
        _RECORDING.keep.extend(t for t in tensors if t is not None)


class _span:
    """Brackets one igemm launch with events on the launch stream when IGEMM_TRACE is a list."""

    def __init__(self, flops, nbytes=0, label=""):
        self.flops = flops
        self.nbytes = nbytes   # algorithmic bytes: every operand and the output once
        self.label = label

    def __enter__(self):
        if IGEMM_TRACE is not None:
            self.s = torch.cuda.Event(enable_timing=True)
            self.e = torch.cuda.Event(enable_timing=True)
            self.s.record()
        return self

    def __exit__(self, *a):
        if IGEMM_TRACE is not None:
            self.e.record()
            IGEMM_TRACE.append((self.s, self.e, self.flops, self.nbytes, self.label))
        return False


TREE_MAX_LEVELS = 12


def _bf16(t, name="tensor"):
    L.require_gpu(t)
    if t.dtype != torch.bfloat16:
        raise RuntimeError("%s must be bfloat16 (got %s)" % (name, t.dtype))
    if not t.is_contiguous():
        raise RuntimeError("%s must be contiguous" % name)
    return t


def _f32(t, name="tensor"):
    L.require_gpu(t)
    if t.dtype != torch.float32:
        raise RuntimeError("%s must be float32 (got %s)" % (name, t.dtype))
    return t.contiguous()


def clip_to_ndhwc(x, in_ntchw, Cp, pads=(0, 0, 0, 0, 0, 0)):
    """f32 (N,C,T,H,W) [or (N,T,C,H,W) if in_ntchw] -> bf16 (N,Tp,Hp,Wp,Cp) with zero halo."""
    x = _f32(x, "clip")
    if in_ntchw:
        N, T, C, H, W = x.shape
    else:
        N, C, T, H, W = x.shape
    pt0, pt1, ph0, ph1, pw0, pw1 = pads
    out = torch.empty((N, T + pt0 + pt1, H + ph0 + ph1, W + pw0 + pw1, Cp), dtype=torch.bfloat16,
                      device=x.device)
    with torch.cuda.device(x.device):
        L.check(L.lib().trace_clip_to_ndhwc_bf16(L.ptr(x), N, C, T, H, W, int(bool(in_ntchw)), Cp, pt0,
                                                 pt1, ph0, ph1, pw0, pw1, L.ptr(out), L.stream()),
                "clip_to_ndhwc")
    return out


def conv3d_stem_f32(x, in_ntchw, w, bias, stride_t, out_thw, relu=True):
    """TRACE's stem (3 -> 64, 5x7x7, pad (2,3,3), stride (stride_t,2,2)) straight from the f32 clip x (N,C,T,H,W) or
    (N,T,C,H,W): no padded bf16 copy of the clip.  w (64, 5*7*8*4) bf16 as for `conv3d_stem`."""
    x = _f32(x, "clip")
    w = _bf16(w, "w")
    if in_ntchw:
        N, T, C, H, W = x.shape
    else:
        N, C, T, H, W = x.shape
    To, Ho, Wo = out_thw
    if w.shape != (64, 5 * 7 * 8 * 4):
        raise RuntimeError("conv3d_stem_f32: weight shape %s" % (tuple(w.shape),))
    out = torch.empty((N, To, Ho, Wo, 64), dtype=torch.bfloat16, device=x.device)
    if bias is not None:
        bias = _f32(bias, "bias")
    with torch.cuda.device(x.device), _span(2.0 * out.numel() * 5 * 7 * 7 * C, 4 * x.numel() + 2 * (w.numel() + out.numel()), "stem_f32 %s" % (tuple(out.shape),)):
        L.check(L.lib().trace_conv3d_stem16_f32(L.ptr(x), N, C, T, H, W, int(bool(in_ntchw)), L.ptr(w), L.ptr(bias),
                                                int(stride_t), To, Ho, Wo, EPI_RELU if relu else 0, L.ptr(out), L.stream()),
                "conv3d_stem_f32")
    return out


def conv3d_stem_u8(x, mean, w, bias, stride_t, out_thw, relu=True):
    """The same stem from uint8 frames x (N,T,H,W,C) channels-last (as decoded): (float)u8 - mean[c] is formed while the
    kernel fills its LDS patch.  mean: C floats (cfg.PIXEL_MEANS) or None.  Bit-identical to `conv3d_stem_f32` on the
    f32 clip holding the same mean-subtracted pixels."""
    import ctypes
    L.require_gpu(x)
    if x.dtype != torch.uint8 or x.dim() != 5 or not x.is_contiguous():
        raise RuntimeError("conv3d_stem_u8: frames must be a contiguous uint8 (N,T,H,W,C) tensor")
    w = _bf16(w, "w")
    N, T, H, W, C = x.shape
    To, Ho, Wo = out_thw
    if w.shape != (64, 5 * 7 * 8 * 4):
        raise RuntimeError("conv3d_stem_u8: weight shape %s" % (tuple(w.shape),))
    out = torch.empty((N, To, Ho, Wo, 64), dtype=torch.bfloat16, device=x.device)
    if bias is not None:
        bias = _f32(bias, "bias")
    mean_c = None
    if mean is not None:
        mv = [float(v) for v in mean]
        if len(mv) != C:
            raise RuntimeError("conv3d_stem_u8: mean needs one value per channel")
        mean_c = (ctypes.c_float * C)(*mv)
    with torch.cuda.device(x.device), _span(2.0 * out.numel() * 5 * 7 * 7 * C, x.numel() + 2 * (w.numel() + out.numel()), "stem_u8 %s" % (tuple(out.shape),)):
        L.check(L.lib().trace_conv3d_stem16_u8(L.ptr(x), N, T, H, W, C, mean_c, L.ptr(w), L.ptr(bias),
                                               int(stride_t), To, Ho, Wo, EPI_RELU if relu else 0, L.ptr(out), L.stream()),
                "conv3d_stem_u8")
    return out


def _out_size(i, k, s, p, d):
    return (i + 2 * p - d * (k - 1) - 1) // s + 1


EPI_POOL_T2 = 128


def conv3d(x, w, bias, ksize, stride=(1, 1, 1), padding=(0, 0, 0), dilation=(1, 1, 1), residual=None,
           relu=False, out_f32=False, pool_t2=False):
    """x (N,T,H,W,Cin) bf16; w (Cout, kt*kh*kw*Cin) bf16 tap-major; bias (Cout) f32 or None.
    pool_t2: also apply MaxPool3d((2,1,1), stride (2,1,1)) to the result - inside the conv's epilogue when it is a
    1x1x1 stride-1 conv over an even number of frames (the frame pair never reaches HBM), as a second launch otherwise."""
    if pool_t2:
        fused = (tuple(ksize) == (1, 1, 1) and tuple(stride) == (1, 1, 1) and tuple(padding) == (0, 0, 0) and
                 x.shape[1] % 2 == 0 and not out_f32 and os.environ.get("TRACE_NO_POOL_FUSE") is None)
        if not fused:
            return maxpool3d(conv3d(x, w, bias, ksize, stride, padding, dilation, residual, relu, out_f32), (2, 1, 1), (2, 1, 1))
    x = _bf16(x, "x")
    w = _bf16(w, "w")
    N, Ti, Hi, Wi, Cin = x.shape
    Cout = w.shape[0]
    kt, kh, kw = ksize
    if w.shape[1] != kt * kh * kw * Cin:
        raise RuntimeError("weight shape %s does not match taps*Cin=%d" % (tuple(w.shape), kt * kh * kw * Cin))
    To = _out_size(Ti, kt, stride[0], padding[0], dilation[0])
    Ho = _out_size(Hi, kh, stride[1], padding[1], dilation[1])
    Wo = _out_size(Wi, kw, stride[2], padding[2], dilation[2])
    epi = (EPI_RELU if relu else 0) | (EPI_RESIDUAL if residual is not None else 0) | (EPI_OUT_F32 if out_f32 else 0)
    if pool_t2:
        epi |= EPI_POOL_T2
    out = torch.empty((N, To // 2 if pool_t2 else To, Ho, Wo, Cout), dtype=torch.float32 if out_f32 else torch.bfloat16,
                      device=x.device)
    if residual is not None:
        residual = _bf16(residual, "residual")
        if tuple(residual.shape) != (N, To, Ho, Wo, Cout):
            raise RuntimeError("residual shape %s != conv output shape %s" % (tuple(residual.shape), (N, To, Ho, Wo, Cout)))
    if bias is not None:
        bias = _f32(bias, "bias")
    ws = workspace(x.device)
    nbytes = 2 * (x.numel() + w.numel()) + out.numel() * out.element_size() + (2 * out.numel() if residual is not None else 0)
    with torch.cuda.device(x.device), _span(2.0 * N * To * Ho * Wo * Cout * kt * kh * kw * Cin, nbytes,
                                            "conv k%s s%s d%s M=%d Cin=%d Cout=%d%s%s" % (tuple(ksize), tuple(stride), tuple(dilation), N * To * Ho * Wo, Cin, Cout,
                                                                                  " +res" if residual is not None else "", " pool" if pool_t2 else "")):
        L.check(L.lib().trace_conv3d_igemm_bf16(
            L.ptr(x), N, Ti, Hi, Wi, Cin, L.ptr(w), L.ptr(bias), Cout, kt, kh, kw, stride[0], stride[1],
            stride[2], padding[0], padding[1], padding[2], dilation[0], dilation[1], dilation[2],
            L.ptr(residual), epi, L.ptr(out), L.ptr(ws), ws.numel(), L.stream()), "conv3d_igemm")
    return out


def conv3d_stem(xp, w, bias, kt, kh, kwp, stride, out_thw, relu=True, alg_k=None):
    """Stem conv on the halo-padded clip xp (N,Tp,Hp,Wp,Cp) bf16; w (Cout, kt*kh*kwp*Cp).
    alg_k = true (unpadded) reduction length kt*kh*kw*Cin, for FLOP accounting."""
    xp = _bf16(xp, "xp")
    w = _bf16(w, "w")
    N, Tp, Hp, Wp, Cp = xp.shape
    Cout = w.shape[0]
    To, Ho, Wo = out_thw
    out = torch.empty((N, To, Ho, Wo, Cout), dtype=torch.bfloat16, device=xp.device)
    if bias is not None:
        bias = _f32(bias, "bias")
    with torch.cuda.device(xp.device), _span(2.0 * out.numel() * (alg_k or kt * kh * kwp * Cp), 2 * (xp.numel() + w.numel() + out.numel())):
        L.check(L.lib().trace_conv3d_stem_bf16(L.ptr(xp), N, Tp, Hp, Wp, Cp, L.ptr(w), L.ptr(bias), Cout,
                                               kt, kh, kwp, stride[0], stride[1], stride[2], To, Ho, Wo,
                                               EPI_RELU if relu else 0, L.ptr(out), L.stream()),
                "conv3d_stem")
    return out


def maxpool3d(x, ksize, stride, padding=(0, 0, 0)):
    x = _bf16(x, "x")
    N, Ti, Hi, Wi, C = x.shape
    To = (Ti + 2 * padding[0] - ksize[0]) // stride[0] + 1
    Ho = (Hi + 2 * padding[1] - ksize[1]) // stride[1] + 1
    Wo = (Wi + 2 * padding[2] - ksize[2]) // stride[2] + 1
    out = torch.empty((N, To, Ho, Wo, C), dtype=torch.bfloat16, device=x.device)
    with torch.cuda.device(x.device):
        L.check(L.lib().trace_maxpool3d_ndhwc_bf16(L.ptr(x), N, Ti, Hi, Wi, C, ksize[0], ksize[1], ksize[2],
                                                   stride[0], stride[1], stride[2], padding[0], padding[1],
                                                   padding[2], L.ptr(out), L.stream()), "maxpool3d")
    return out


EPI_LEAKY = 8
EPI_RES_POST = 16


def _rows(t, name):
    """(batch, rows, ld, batch_stride) of a 2-D/3-D bf16 operand whose last dim is contiguous."""
    L.require_gpu(t)
    if t.dtype != torch.bfloat16:
        raise RuntimeError("%s must be bfloat16" % name)
    if t.stride(-1) != 1:
        raise RuntimeError("%s: last dimension must be contiguous" % name)
    if t.dim() == 2:
        return 1, t.shape[0], t.stride(0), 0
    if t.dim() == 3:
        return t.shape[0], t.shape[1], t.stride(1), t.stride(0)
    raise RuntimeError("%s must be 2-D or 3-D" % name)


EPI_RES_F32 = 32
_ACT = {None: 0, "none": 0, "relu": EPI_RELU, "leaky": EPI_LEAKY}


EPI_WTILED = 64


def tile_weights(w):
    """(N, K) bf16 row-major -> the pre-tiled layout of TRACE_EPI_WTILED (N % 128 == 0, K % 64 == 0): same
    shape / bytes, blocks [N/128][K/64] of 128 rows x 64 k with the 16-byte chunks of row r stored at chunk
    position pc holding logical chunk pc ^ a_key(r), a_key(r) = bits (0, 1, 3) of r (csrc/igemm.hip: the swizzle that
    keeps the weight fragment reads free of LDS bank conflicts)."""
    N, K = w.shape
    if N % 128 or K % 64 or w.dtype != torch.bfloat16:
        raise RuntimeError("tile_weights: needs bf16 (N % 128 == 0, K % 64 == 0)")
    t = w.view(N // 128, 128, K // 64, 8, 8).permute(0, 2, 1, 3, 4)          # [nt][kt][r][chunk][8]
    r = torch.arange(128, device=w.device).view(128, 1)
    pc = torch.arange(8, device=w.device).view(1, 8)
    src = (pc ^ ((r & 3) | ((r >> 1) & 4))).view(1, 1, 128, 8, 1).expand(N // 128, K // 64, 128, 8, 8)
    return torch.gather(t, 3, src).contiguous().view(N, K)


def gemm_nt_dual(A, A2, B, bias=None, relu=False, out_f32=False, out2=None):
    """relu?([A | A2] @ B^T + bias): A (M,K1), A2 (M,K2) bf16 row-strided views, B (N, K1+K2) bf16; K1, K2 % 64 == 0.
    One launch for relu(conv3(h) + downsample(x)) of a bottleneck with a stride-1 downsample.
    out_f32 / out2 as in `gemm_nt` (f32 result plus its bf16 copy: the f32 residual stream of the late stages)."""
    _, M, lda, _ = _rows(A, "A")
    _, M2, lda2, _ = _rows(A2, "A2")
    _, N, ldb, _ = _rows(B, "B")
    K1, K2 = A.shape[-1], A2.shape[-1]
    if A.dim() != 2 or A2.dim() != 2 or B.dim() != 2 or M2 != M or B.shape[1] != K1 + K2:
        raise RuntimeError("gemm_nt_dual: shape mismatch")
    out = torch.empty((M, N), dtype=torch.float32 if out_f32 else torch.bfloat16, device=A.device)
    c2 = torch.empty((M, N), dtype=torch.bfloat16, device=A.device) if out2 is not None else None
    if bias is not None:
        bias = _f32(bias, "bias")
    epi = (EPI_RELU if relu else 0) | (EPI_OUT_F32 if out_f32 else 0)
    nbytes = 2 * (M * (K1 + K2) + N * (K1 + K2)) + out.numel() * out.element_size() + (2 * M * N if c2 is not None else 0)
    with torch.cuda.device(A.device), _span(2.0 * M * N * (K1 + K2), nbytes, "gemm_dual M=%d N=%d K=%d+%d" % (M, N, K1, K2)):
        L.check(L.lib().trace_gemm_bf16_nt_dual_ex(L.ptr(A), lda, K1, L.ptr(A2), lda2, K2, L.ptr(B), ldb, M, N, N, L.ptr(bias),
                                                   None, epi, L.ptr(out), L.ptr(c2), _ACT[out2] if out2 is not None else 0,
                                                   L.stream()), "gemm_bf16_nt_dual")
    return out if out2 is None else (out, c2)


def gemm_nt(A, B, bias=None, residual=None, relu=False, alpha=1.0, out_f32=False, leaky=False,
            residual_post=False, out=None, out2=None, b_tiled=False):
    """C = alpha * A @ B^T (+bias)(+residual)(+ReLU | LeakyReLU(0.1))(+residual after the activation).
    A (M,K) or (b,M,K); B (N,K) or (b,N,K); row-strided views are accepted (last dim contiguous).
    `out` may be a row-strided view (e.g. a column slice of a wider zero-initialised buffer).
    residual: bf16 or f32 tensor of the output's shape.  bias: (N) or, batched, (b,N).
    out2: None, or "none" | "relu" | "leaky": also return bf16(act(C)) (the next Linear's operand) -> (C, C2)."""
    ba, M, lda, sA = _rows(A, "A")
    bb, N, ldb, sB = _rows(B, "B")
    K = A.shape[-1]
    if B.shape[-1] != K:
        raise RuntimeError("gemm_nt: K mismatch %s vs %s" % (tuple(A.shape), tuple(B.shape)))
    b = max(ba, bb)
    if ba not in (1, b) or bb not in (1, b):
        raise RuntimeError("gemm_nt: batch mismatch")
    if A.dim() == 3 and ba == 1:
        sA = 0
    out_shape = (b, M, N) if (A.dim() == 3 or B.dim() == 3) else (M, N)
    odt = torch.float32 if out_f32 else torch.bfloat16
    if out is None:
        out = torch.empty(out_shape, dtype=odt, device=A.device)
    else:
        if tuple(out.shape) != out_shape or out.dtype != odt or out.stride(-1) != 1:
            raise RuntimeError("gemm_nt: bad `out`")
    ldc = out.stride(-2)
    sC = out.stride(0) if out.dim() == 3 else 0
    epi = (EPI_RELU if relu else 0) | (EPI_OUT_F32 if out_f32 else 0) | (EPI_LEAKY if leaky else 0)
    if b_tiled:
        if B.dim() != 2 or not B.is_contiguous() or A.dim() != 2:
            raise RuntimeError("gemm_nt: b_tiled takes one contiguous pre-tiled weight matrix (ops.tile_weights)")
        epi |= EPI_WTILED
    if residual is not None:
        L.require_gpu(residual)
        if residual.dtype == torch.float32:
            epi |= EPI_RES_F32
        elif residual.dtype != torch.bfloat16:
            raise RuntimeError("gemm_nt: residual must be bf16 or f32")
        if tuple(residual.shape) != out_shape or not out.is_contiguous() or not residual.is_contiguous():
            raise RuntimeError("gemm_nt: residual must match a contiguous output")
        epi |= EPI_RES_POST if residual_post else EPI_RESIDUAL
    sBias = 0
    if bias is not None:
        bias = _f32(bias, "bias")
        if bias.dim() == 2:
            if bias.shape[0] != b:
                raise RuntimeError("gemm_nt: batched bias must be (b, N)")
            sBias = bias.stride(0)
    c2 = None
    if out2 is not None:
        if out2 not in _ACT:
            raise RuntimeError("gemm_nt: out2 must be one of none/relu/leaky")
        if not out.is_contiguous():
            raise RuntimeError("gemm_nt: out2 needs a contiguous output")
        c2 = torch.empty(out_shape, dtype=torch.bfloat16, device=A.device)
    ws = workspace(A.device)
    nbytes = (2 * (ba * M * K + bb * N * K) + out.numel() * out.element_size() + (2 * c2.numel() if c2 is not None else 0) +
              (residual.numel() * residual.element_size() if residual is not None else 0))
    with torch.cuda.device(A.device), _span(2.0 * b * M * N * K, nbytes, "gemm b=%d M=%d N=%d K=%d%s" % (b, M, N, K, " tiled" if b_tiled else "")):
        L.check(L.lib().trace_gemm_bf16_nt_ex(L.ptr(A), L.ptr(B), M, N, K, lda, ldb, ldc, b, sA, sB, sC, float(alpha),
                                              L.ptr(bias), sBias, L.ptr(residual), epi, L.ptr(out), L.ptr(c2),
                                              _ACT[out2], L.ptr(ws), ws.numel(), L.stream()),
                "gemm_bf16_nt")
    return out if out2 is None else (out, c2)


def transpose(x, pad_to=1):
    """(b, R, C) bf16 -> (b, C, Rp) bf16 with Rp = R rounded up to `pad_to` (zero filled); returns the
    full padded tensor (2-D input -> 2-D output)."""
    x = _bf16(x, "x")
    two_d = x.dim() == 2
    if two_d:
        x = x.unsqueeze(0)
    b, R, C = x.shape
    Rp = (R + pad_to - 1) // pad_to * pad_to
    out = (torch.zeros if Rp != R else torch.empty)((b, C, Rp), dtype=torch.bfloat16, device=x.device)
    with torch.cuda.device(x.device):
        L.check(L.lib().trace_transpose_bf16(L.ptr(x), b, R, C, L.ptr(out), Rp, C * Rp, L.stream()), "transpose")
    return out[0] if two_d else out


def nl_pool_split(tpg, c, transposed_phi=False):
    """tpg (N,T,H,W,3c) bf16 (theta|phi|g) -> phi_p (N,nkp,c), g^T (N,c,nkp): (1,2,2) max-pool of the phi / g
    thirds in the operand layouts of the non-local block's two products; nkp = nkv rounded up to 8.
    transposed_phi: return phi^T (N,c,nkp) instead of phi_p (operands of the reassociated product phi^T g)."""
    tpg = _bf16(tpg, "tpg")
    N, T, H, W, C3 = tpg.shape
    if C3 != 3 * c:
        raise RuntimeError("nl_pool_split: last dim %d != 3*%d" % (C3, c))
    nkv = T * (H // 2) * (W // 2)
    nkp = (nkv + 7) // 8 * 8
    phi = torch.empty((N, c, nkp) if transposed_phi else (N, nkp, c), dtype=torch.bfloat16, device=tpg.device)
    gT = torch.empty((N, c, nkp), dtype=torch.bfloat16, device=tpg.device)
    with torch.cuda.device(tpg.device):
        L.check(L.lib().trace_nl_pool_split2_bf16(L.ptr(tpg), N, T, H, W, c, None if transposed_phi else L.ptr(phi),
                                                  L.ptr(phi) if transposed_phi else None, L.ptr(gT), nkp, L.stream()),
                "nl_pool_split")
    return phi, gT, nkv


def roi_align_nhwc(x, rois, spatial_scale, pooled, sampling_ratio):
    """x (B,H,W,C) bf16, rois (K,5) f32 -> (K, pooled, pooled, C) bf16."""
    x = _bf16(x, "x")
    rois = _f32(rois, "rois")
    B, H, W, C = x.shape
    K = rois.shape[0]
    out = torch.empty((K, pooled, pooled, C), dtype=torch.bfloat16, device=x.device)
    if K == 0:
        return out
    with torch.cuda.device(x.device):
        L.check(L.lib().trace_roi_align_fwd_nhwc_bf16(L.ptr(x), B, H, W, C, L.ptr(rois), K,
                                                      float(spatial_scale), pooled, pooled,
                                                      int(sampling_ratio), L.ptr(out), L.stream()),
                "roi_align_nhwc")
    return out


def roi_align_avg_nhwc(x, rois, spatial_scale, pooled, sampling_ratio):
    """ROIAlign + mean over the pooled bins: x (B,H,W,C) bf16, rois (K,5) f32 -> (K,C) bf16."""
    x = _bf16(x, "x")
    rois = _f32(rois, "rois")
    B, H, W, C = x.shape
    K = rois.shape[0]
    out = torch.empty((K, C), dtype=torch.bfloat16, device=x.device)
    if K == 0:
        return out
    with torch.cuda.device(x.device):
        L.check(L.lib().trace_roi_align_avg_nhwc_bf16(L.ptr(x), B, H, W, C, L.ptr(rois), K,
                                                      float(spatial_scale), pooled, pooled,
                                                      int(sampling_ratio), L.ptr(out), L.stream()),
                "roi_align_avg_nhwc")
    return out


def temporal_attention(qs, qo, o, inv_sqrt_d, out_bf16=False):
    """qs (n,D), qo (n,T,D), o (n,T,K) f32 -> (n,K) f32, or bf16 with out_bf16 (cluster_tree3d.py:337-366)."""
    qs, qo, o = _f32(qs), _f32(qo), _f32(o)
    n, T, D = qo.shape
    K = o.shape[2]
    out = torch.empty((n, K), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=o.device)
    with torch.cuda.device(o.device):
        L.check(L.lib().trace_temporal_attention_ex(L.ptr(qs), L.ptr(qo), L.ptr(o), n, T, D, K, float(inv_sqrt_d),
                                                    None if out_bf16 else L.ptr(out), L.ptr(out) if out_bf16 else None,
                                                    L.stream()), "temporal_attention")
    return out


def pair_attention(kmap, vmap, rel_rois, sbj_inds, obj_inds, q_s, q_o, spatial_scale, pooled=7):
    """kmap, vmap (B,H,W,D) bf16; returns s_map, o_map (views of one (P, pooled*pooled, 2D) buffer: the
    concatenated operand of the A|B 1x1-conv GEMM) and v_map (P, pooled*pooled, D) bf16."""
    kmap, vmap = _bf16(kmap, "kmap"), _bf16(vmap, "vmap")
    rel_rois, q_s, q_o = _f32(rel_rois), _f32(q_s), _f32(q_o)
    L.require_gpu(sbj_inds, obj_inds)
    sbj_inds = sbj_inds.to(torch.int32).contiguous()
    obj_inds = obj_inds.to(torch.int32).contiguous()
    B, H, W, D = kmap.shape
    P = rel_rois.shape[0]
    so = torch.empty((P, pooled * pooled, 2 * D), dtype=torch.bfloat16, device=kmap.device)
    v = torch.empty((P, pooled * pooled, D), dtype=torch.bfloat16, device=kmap.device)
    s_map, o_map = so[:, :, :D], so[:, :, D:]
    with torch.cuda.device(kmap.device):
        L.check(L.lib().trace_pair_attention_fused(L.ptr(kmap), L.ptr(vmap), B, H, W, D, L.ptr(rel_rois),
                                                   L.ptr(sbj_inds), L.ptr(obj_inds), P, L.ptr(q_s), L.ptr(q_o),
                                                   float(spatial_scale), pooled, 2 * D, L.ptr(s_map),
                                                   L.ptr(o_map), L.ptr(v), L.stream()),
                "pair_attention")
    return s_map, o_map, v, so


def pair_infer(vab, qk, rel_rois, sbj_inds, obj_inds, bias, spatial_scale, pooled=7, partner=None):
    """vab (B,H,W,3D) bf16 = [V | W_A V | W_B V]; qk (2*n_obj, >= B*H*W) f32 (row 2i = q_s of object i, 2i+1 = q_o);
    bias (D) f32 = b_A + b_B.  Returns sp (P, pooled*pooled, D) bf16 = LeakyReLU(A(s_map) + B(o_map)) + V
    (model_builder_rel.py:1039-1040) without forming s_map / o_map.  partner (P) int32, optional: index of the pair
    with subject and object swapped (-1: none; see `swapped_pair_index`): the two share one union box, so it is
    pooled once for both."""
    vab = _bf16(vab, "vab")
    rel_rois, bias = _f32(rel_rois), _f32(bias)
    L.require_gpu(sbj_inds, obj_inds, qk)
    if qk.dtype != torch.float32 or qk.dim() != 2 or qk.stride(1) != 1:
        raise RuntimeError("pair_infer: qk must be 2-D f32 with contiguous rows")
    if sbj_inds.dtype != torch.int32 or obj_inds.dtype != torch.int32:
        raise RuntimeError("pair_infer: indices must be int32")
    B, H, W, D3 = vab.shape
    D = D3 // 3
    P = rel_rois.shape[0]
    if qk.shape[1] < B * H * W or bias.numel() != D:
        raise RuntimeError("pair_infer: qk / bias shape mismatch")
    sp = torch.empty((P, pooled * pooled, D), dtype=torch.bfloat16, device=vab.device)
    with torch.cuda.device(vab.device):
        if partner is not None:
            L.require_gpu(partner)
            if partner.dtype != torch.int32 or partner.numel() != P or not partner.is_contiguous():
                raise RuntimeError("pair_infer: partner must be a contiguous int32 tensor with one entry per pair")
        L.check(L.lib().trace_pair_infer_fused(L.ptr(vab), B, H, W, D, L.ptr(qk), qk.stride(0), L.ptr(rel_rois),
                                               L.ptr(sbj_inds.contiguous()), L.ptr(obj_inds.contiguous()),
                                               None if partner is None else L.ptr(partner), P, L.ptr(bias),
                                               float(spatial_scale), pooled, L.ptr(sp), L.stream()), "pair_infer")
    return sp


PAIR_MFMA_PITCH = 224


def pair_mfma_supported(H, W, pooled, D):
    """Shapes trace_pair_infer_mfma takes: maps of at most 16 x 16 pixels (H*W + 31 <= the 224-pixel pitch), pooled <= 8."""
    return H <= 16 and W <= 16 and pooled <= 8 and (H * W + 7) // 8 * 8 + 24 <= PAIR_MFMA_PITCH and D % 32 == 0 and D <= 512


def pair_maps_channel_major(fmap, w_vab):
    """fmap (B,H,W,C) bf16, w_vab (3D, C) bf16 -> vabT (B, 3D, 224) bf16 = w_vab fmap^T per clip (NO bias), pixel axis
    contiguous and zero-padded: the operand layout of `pair_infer_mfma`."""
    fmap = _bf16(fmap, "fmap")
    B, H, W, C = fmap.shape
    vabT = torch.zeros((B, w_vab.shape[0], PAIR_MFMA_PITCH), dtype=torch.bfloat16, device=fmap.device)
    gemm_nt(w_vab, fmap.view(B, H * W, C), out=vabT[:, :, :H * W])
    return vabT


def pair_work_records(X, Y, union, partner):
    """Host side (numpy): the work list of `pair_infer_mfma` - one 12-word record per union box to pool: [pair p, swapped
    pair q or -1, X[p], Y[p], union[p] (5 floats, bit patterns), 0, 0, 0], for every pair that has no swapped partner
    or is the lower index of one."""
    import numpy as np
    X, Y = np.asarray(X, np.int32).reshape(-1), np.asarray(Y, np.int32).reshape(-1)
    part = np.asarray(partner, np.int32).reshape(-1) if partner is not None else np.full(X.shape[0], -1, np.int32)
    lead = np.nonzero((part < 0) | (part > np.arange(X.shape[0])))[0]
    rec = np.zeros((lead.shape[0], 12), np.int32)
    rec[:, 0], rec[:, 1], rec[:, 2], rec[:, 3] = lead, part[lead], X[lead], Y[lead]
    rec[:, 4:9] = np.ascontiguousarray(np.asarray(union, np.float32)[lead]).view(np.int32)
    return rec


def pair_infer_mfma(vabT, map_hw, vbias, qk, rel_rois, sbj_inds, obj_inds, bias, spatial_scale, pooled=7, partner=None,
                    work=None):
    """`pair_infer` on the matrix cores: vabT (B,3D,224) bf16 from `pair_maps_channel_major`, map_hw = (H, W), vbias (3D)
    f32 = the bias of the map GEMM (or None).  Same result as pair_infer up to bf16 rounding of the pooling weights.
    work: (n, 12) int32 device tensor of `pair_work_records` (then rel_rois / indices / partner are only used for P)."""
    vabT = _bf16(vabT, "vabT")
    rel_rois, bias = _f32(rel_rois), _f32(bias)
    L.require_gpu(sbj_inds, obj_inds, qk)
    if qk.dtype != torch.float32 or qk.dim() != 2 or qk.stride(1) != 1:
        raise RuntimeError("pair_infer_mfma: qk must be 2-D f32 with contiguous rows")
    if sbj_inds.dtype != torch.int32 or obj_inds.dtype != torch.int32:
        raise RuntimeError("pair_infer_mfma: indices must be int32")
    B, D3, ldp = vabT.shape
    D = D3 // 3
    H, W = map_hw
    P = rel_rois.shape[0]
    if qk.shape[1] < B * H * W or bias.numel() != D or ldp != PAIR_MFMA_PITCH:
        raise RuntimeError("pair_infer_mfma: qk / bias / pitch mismatch")
    if vbias is not None:
        vbias = _f32(vbias)
        if vbias.numel() != D3:
            raise RuntimeError("pair_infer_mfma: vbias must hold 3D values")
    if work is not None:
        L.require_gpu(work)
        if work.dtype != torch.int32 or work.dim() != 2 or work.shape[1] != 12 or not work.is_contiguous():
            raise RuntimeError("pair_infer_mfma: work must be a contiguous (n, 12) int32 tensor (ops.pair_work_records)")
    sp = torch.empty((P, pooled * pooled, D), dtype=torch.bfloat16, device=vabT.device)
    with torch.cuda.device(vabT.device):
        L.check(L.lib().trace_pair_infer_mfma(L.ptr(vabT), B, H, W, D, ldp, L.ptr(vbias), L.ptr(qk), qk.stride(0),
                                              L.ptr(rel_rois), L.ptr(sbj_inds.contiguous()), L.ptr(obj_inds.contiguous()),
                                              _partner_ptr(partner, P, "pair_infer_mfma"), P,
                                              None if work is None else L.ptr(work), 0 if work is None else int(work.shape[0]),
                                              L.ptr(bias), float(spatial_scale), pooled, L.ptr(sp), L.stream()), "pair_infer_mfma")
    return sp


def _partner_ptr(partner, P, what):
    if partner is None:
        return None
    L.require_gpu(partner)
    if partner.dtype != torch.int32 or partner.numel() != P or not partner.is_contiguous():
        raise RuntimeError(what + ": partner must be a contiguous int32 tensor with one entry per pair")
    return L.ptr(partner)


def pair_pool(vab, rel_rois, spatial_scale, pooled=7, partner=None):
    """Union-box pooling of vab (B,H,W,3D) bf16 -> (P, pooled*pooled, 3D) bf16 (first half of `pair_infer`).  With
    `partner` only the lower index of a swapped pair is pooled (the other row is left unwritten)."""
    vab = _bf16(vab, "vab")
    rel_rois = _f32(rel_rois)
    B, H, W, D3 = vab.shape
    P = rel_rois.shape[0]
    out = torch.empty((P, pooled * pooled, D3), dtype=torch.bfloat16, device=vab.device)
    with torch.cuda.device(vab.device):
        L.check(L.lib().trace_pair_pool(L.ptr(vab), B, H, W, D3 // 3, L.ptr(rel_rois), _partner_ptr(partner, P, "pair_pool"),
                                        P, float(spatial_scale), pooled, L.ptr(out), L.stream()), "pair_pool")
    return out


def pair_combine(pooled_maps, map_shape, qk, rel_rois, sbj_inds, obj_inds, bias, spatial_scale, pooled=7, partner=None):
    """Second half of `pair_infer`: pooled_maps (P, pooled*pooled, 3D) from `pair_pool` (called with the same
    `partner`), map_shape = (B, H, W)."""
    pooled_maps = _bf16(pooled_maps, "pooled_maps")
    rel_rois, bias = _f32(rel_rois), _f32(bias)
    L.require_gpu(sbj_inds, obj_inds, qk)
    if qk.dtype != torch.float32 or qk.dim() != 2 or qk.stride(1) != 1:
        raise RuntimeError("pair_combine: qk must be 2-D f32 with contiguous rows")
    if sbj_inds.dtype != torch.int32 or obj_inds.dtype != torch.int32:
        raise RuntimeError("pair_combine: indices must be int32")
    B, H, W = map_shape
    P, cells, D3 = pooled_maps.shape
    D = D3 // 3
    if qk.shape[1] < B * H * W or bias.numel() != D or cells != pooled * pooled or rel_rois.shape[0] != P:
        raise RuntimeError("pair_combine: shape mismatch")
    sp = torch.empty((P, cells, D), dtype=torch.bfloat16, device=pooled_maps.device)
    with torch.cuda.device(pooled_maps.device):
        L.check(L.lib().trace_pair_combine(L.ptr(pooled_maps), B, H, W, D, L.ptr(qk), qk.stride(0), L.ptr(rel_rois),
                                           L.ptr(sbj_inds.contiguous()), L.ptr(obj_inds.contiguous()),
                                           _partner_ptr(partner, P, "pair_combine"), P, L.ptr(bias),
                                           float(spatial_scale), pooled, L.ptr(sp), L.stream()), "pair_combine")
    return sp


def tree_gru(x, F, Hd, n_groups, forest, weights, max_tree_nodes=0, wide=True):
    """x (n_nodes, n_groups*F) f32; forest = dict(tree_off, level_off, n_levels, parent, child_off,
    child_idx) of int32 device tensors; weights = list (per group) of dicts holding the 16 TRANSPOSED
    f32 device tensors named as in trace_tree_gru_weights.  max_tree_nodes: host-known upper bound of the
    nodes of one tree (enables the LDS-resident kernel).  wide=False: the one-plane workspace of trace_tree_gru
    (never the batched-product kernels for matrices beyond LDS).  Returns h, hd (n_nodes, n_groups*Hd)."""
    x = _f32(x)
    n_nodes = x.shape[0]
    dev = x.device
    h = torch.empty((n_nodes, n_groups * Hd), dtype=torch.float32, device=dev)
    hd = torch.empty_like(h)
    ws_bytes = int(L.lib().trace_tree_gru_workspace_bytes(n_nodes, F, Hd, n_groups)) if wide else h.numel() * 4
    ws = torch.empty((max(ws_bytes, 4) + 3) // 4, dtype=torch.float32, device=dev)
    arr = (L.TreeGruWeights * n_groups)()
    keep = []
    for g in range(n_groups):
        for name, _ in L.TreeGruWeights._fields_:
            t = _f32(weights[g][name])
            keep.append(t)
            setattr(arr[g], name, t.data_ptr())
    n_trees = forest["tree_off"].numel() - 1
    with torch.cuda.device(dev):
        L.check(L.lib().trace_tree_gru_ex(L.ptr(x), n_nodes, F, Hd, n_groups, n_trees, L.ptr(forest["tree_off"]),
                                          L.ptr(forest["level_off"]), L.ptr(forest["n_levels"]),
                                          L.ptr(forest["parent"]), L.ptr(forest["child_off"]),
                                          L.ptr(forest["child_idx"]), arr, int(max_tree_nodes), L.ptr(h), L.ptr(hd), L.ptr(ws),
                                          ws.numel() * 4, L.stream()), "tree_gru")
    return h, hd


RANK_MAX_PRD = 256
RANK_MAX_CAND = 16384


def rank_triplets_supported(n_prd, max_pairs_per_clip, prd_k, topk=100):
    """One on-chip sort per clip up to RANK_MAX_CAND candidates; above that segments of RANK_MAX_CAND whose topk
    survivors are merged on chip (segments x topk <= RANK_MAX_CAND)."""
    cand = max_pairs_per_clip * min(prd_k, n_prd)
    n_seg = max(1, -(-cand // RANK_MAX_CAND))
    return n_prd <= RANK_MAX_PRD and (n_seg == 1 or n_seg * topk <= RANK_MAX_CAND)


def rank_triplets(prd, sbj_scores, obj_scores, sbj_labels, obj_labels, pair_off, max_pairs_per_clip, topk, prd_k=1):
    """task_evaluation_vg_and_vrd.py:207-228 for every clip at once.  prd (P,n_prd) f32; scores (P) f32; labels
    (P) int64; pair_off (B+1) int64 on the GPU.  Returns dict of (B, topk) tensors (rows padded with score -1)."""
    sbj_scores, obj_scores = _f32(sbj_scores), _f32(obj_scores)
    L.require_gpu(prd, sbj_labels, obj_labels, pair_off)
    if prd.dtype != torch.float32 or prd.dim() != 2 or prd.stride(1) != 1:
        raise RuntimeError("rank_triplets: prd must be (P, n_prd) f32 with contiguous rows")
    if sbj_labels.dtype != torch.int64 or obj_labels.dtype != torch.int64 or pair_off.dtype != torch.int64:
        raise RuntimeError("rank_triplets: labels and pair_off must be int64")
    sbj_labels, obj_labels, pair_off = sbj_labels.contiguous(), obj_labels.contiguous(), pair_off.contiguous()
    P, n_prd = prd.shape
    B = pair_off.numel() - 1
    dev = prd.device
    kk = min(prd_k, n_prd)
    lib = L.lib()
    ws_bytes = lib.trace_rank_triplets_workspace_bytes_ex(P, kk, B, int(max_pairs_per_clip), int(topk))
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    top = torch.empty((B, topk), dtype=torch.float32, device=dev)
    outs = [torch.empty((B, topk), dtype=torch.int64, device=dev) for _ in range(4)]
    with torch.cuda.device(dev):
        L.check(lib.trace_rank_triplets(L.ptr(prd), P, n_prd, prd.stride(0), L.ptr(sbj_scores), L.ptr(obj_scores), L.ptr(sbj_labels),
                                        L.ptr(obj_labels), L.ptr(pair_off), B, int(max_pairs_per_clip), int(prd_k),
                                        int(topk), L.ptr(top), L.ptr(outs[0]), L.ptr(outs[1]), L.ptr(outs[2]),
                                        L.ptr(outs[3]), L.ptr(ws), ws_bytes, L.stream()), "rank_triplets")
    return dict(top_scores=top, top_pair=outs[0], top_prd=outs[1], top_sbj=outs[2], top_obj=outs[3])


def upload_slots(pinned, slot_bytes, counter, dst, nbytes):
    """First node of a captured step: dst[:nbytes] = slot (launch count & 1) of the two-slot PINNED host
    buffer (see trace_upload_slots); `counter` is a zero-initialised 2-element int32 device tensor owned by the graph."""
    L.require_gpu(counter, dst)
    if pinned.is_cuda or not pinned.is_pinned():
        raise RuntimeError("upload_slots: `pinned` must be pinned host memory")
    if counter.numel() < 2 or counter.dtype != torch.int32:
        raise RuntimeError("upload_slots: counter must hold two int32 words")
    if pinned.numel() < 2 * slot_bytes or dst.numel() < nbytes:
        raise RuntimeError("upload_slots: buffers too small")
    with torch.cuda.device(dst.device):
        L.check(L.lib().trace_upload_slots(ctypes.c_void_p(pinned.data_ptr()), slot_bytes, L.ptr(counter), L.ptr(dst),
                                           nbytes, L.stream()), "upload_slots")


def act_cast(a, act=None, b=None):
    """bf16(act([a | b])): a (rows, wa) f32, optional b (rows, wb) f32; act None | "relu" | "leaky"."""
    a = _f32(a)
    rows, wa = a.shape
    wb = 0
    if b is not None:
        b = _f32(b)
        if b.shape[0] != rows:
            raise RuntimeError("act_cast: row mismatch")
        wb = b.shape[1]
    out = torch.empty((rows, wa + wb), dtype=torch.bfloat16, device=a.device)
    with torch.cuda.device(a.device):
        L.check(L.lib().trace_act_cast_bf16(L.ptr(a), wa, L.ptr(b), wb, rows, _ACT[act], L.ptr(out), L.stream()), "act_cast")
    return out


def layernorm_act(x, gamma, beta, eps, act="relu"):
    """bf16(act(LayerNorm(x))) over the last dim: x (rows, w) f32."""
    x = _f32(x)
    rows, w = x.shape
    gamma = _f32(gamma) if gamma is not None else None
    beta = _f32(beta) if beta is not None else None
    out = torch.empty((rows, w), dtype=torch.bfloat16, device=x.device)
    with torch.cuda.device(x.device):
        L.check(L.lib().trace_layernorm_act_bf16(L.ptr(x), rows, w, L.ptr(gamma), L.ptr(beta), float(eps), _ACT[act],
                                                 L.ptr(out), L.stream()), "layernorm_act")
    return out


def gather_add_act(a, b, idx, act=None):
    """bf16(act(a + b[idx])): a (rows,w) f32, b (*,w) f32, idx (rows) int64."""
    a, b = _f32(a), _f32(b)
    L.require_gpu(idx)
    if idx.dtype != torch.int64 or a.shape[1] != b.shape[1] or idx.numel() != a.shape[0]:
        raise RuntimeError("gather_add_act: bad arguments")
    out = torch.empty(tuple(a.shape), dtype=torch.bfloat16, device=a.device)
    with torch.cuda.device(a.device):
        L.check(L.lib().trace_gather_add_act_bf16(L.ptr(a), L.ptr(b), L.ptr(idx.contiguous()), a.shape[0], a.shape[1],
                                                  _ACT[act], L.ptr(out), L.stream()), "gather_add_act")
    return out


def pair_assemble(det_labels, det_scores, X, Y, low_s, low_o, part_s, part_o, freq, n_rel, n_obj, per_pair=None):
    """Per-pair gathers (relpn_heads.py:113-116, model_builder_rel.py:1064-1075, 1662): returns sbj_lab, obj_lab
    (P) int64, sbj_sc, obj_sc (P) f32, so_bias = low_s[X]+low_o[Y] (P,wl) f32, so_part = part_s[X]+part_o[Y]
    (+ freq[(ls-1)*n_obj + lo-1]) (P,wp) f32.  per_pair = (sbj_lab, obj_lab int64, sbj_sc, obj_sc f32), one entry per
    pair: labels / scores given per pair (`use_gt_labels`, model_builder_rel.py:844-858) instead of gathered per roi."""
    det_scores, low_s, low_o, part_s, part_o = _f32(det_scores), _f32(low_s), _f32(low_o), _f32(part_s), _f32(part_o)
    L.require_gpu(det_labels, X, Y)
    if det_labels.dtype != torch.int64 or X.dtype != torch.int64 or Y.dtype != torch.int64:
        raise RuntimeError("pair_assemble: labels and indices must be int64")
    if freq is not None:
        freq = _f32(freq)
        if freq.shape[1] != n_rel:
            raise RuntimeError("pair_assemble: frequency table width != n_rel")
    P = X.numel()
    wl, wp = low_s.shape[1], part_s.shape[1]
    dev = low_s.device
    sl = torch.empty((P,), dtype=torch.int64, device=dev)
    ol = torch.empty((P,), dtype=torch.int64, device=dev)
    ss = torch.empty((P,), dtype=torch.float32, device=dev)
    os_ = torch.empty((P,), dtype=torch.float32, device=dev)
    so_bias = torch.empty((P, wl), dtype=torch.float32, device=dev)
    so_part = torch.empty((P, wp), dtype=torch.float32, device=dev)
    pp = [None] * 4
    if per_pair is not None:
        L.require_gpu(*per_pair)
        if (per_pair[0].dtype != torch.int64 or per_pair[1].dtype != torch.int64 or per_pair[2].dtype != torch.float32 or
                per_pair[3].dtype != torch.float32 or any(t.numel() != P for t in per_pair)):
            raise RuntimeError("pair_assemble: per_pair = (int64, int64, f32, f32), one entry per pair")
        pp = [t.contiguous() for t in per_pair]
    with torch.cuda.device(dev):
        L.check(L.lib().trace_pair_assemble_ex(L.ptr(det_labels.contiguous()), L.ptr(det_scores), L.ptr(X.contiguous()),
                                               L.ptr(Y.contiguous()), P, L.ptr(pp[0]), L.ptr(pp[1]), L.ptr(pp[2]), L.ptr(pp[3]),
                                               L.ptr(low_s), L.ptr(low_o), wl, L.ptr(part_s),
                                               L.ptr(part_o), wp, L.ptr(freq), int(n_rel), int(n_obj), L.ptr(sl), L.ptr(ol),
                                               L.ptr(ss), L.ptr(os_), L.ptr(so_bias), L.ptr(so_part), L.stream()),
                "pair_assemble")
    return sl, ol, ss, os_, so_bias, so_part
