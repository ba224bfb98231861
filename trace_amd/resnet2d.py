"""2-D ResNet conv bodies of TRACE through the implicit-GEMM conv family (a frame is a one-frame clip, kt = 1).

Reference: Detectron_pytorch/lib/modeling/ResNet.py - `ResNet_convX_body` (:49-123), `add_stage` /
`add_residual_block` (:166-198), `basic_bn_shortcut` (:205-213), `basic_bn_stem` (:232-238),
`bottleneck_transformation` (:254-301) - and the frozen-BN `AffineChannel2d` (nn/modules/affine.py:6-37).
TRACE builds two of them (lib/modeling_rel/model_builder_rel.py:116, :188-190): `Conv_Body` =
ResNet101_conv5_body on the detection key frame and `Prd_RCNN2D` = ResNet50_conv5_body_dilation2 on the relation
frame, both with RESNETS.STRIDE_1X1 False and RES5_DILATION 2 in the VidVRD / AG configs (stride on the 3x3,
res5 dilated, output stride 16).

Module and buffer names are the reference's (res1.conv1, res1.bn1, res2.0.conv1, res2.0.bn1, res2.0.downsample.0/.1
...), so its checkpoints load with load_state_dict.  Inference only: AffineChannel2d is folded into the conv weights
(scale) and the epilogue bias; ReLU and the residual add ride in the conv epilogue.
"""
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .resnet3d import _PACK_EPOCH


class AffineChannel2d(nn.Module):
    """Frozen batch norm (affine.py:6-37): buffers only, y = (x - mean) * weight / sqrt(var + eps) + bias."""

    def __init__(self, num_features, eps=1e-5):
        super().__init__()
        self.num_features = num_features
        self.eps = eps
        self.register_buffer("weight", torch.ones(num_features))
        self.register_buffer("bias", torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features) - eps)

    def scale_shift(self):
        scale = self.weight.float() * (self.running_var.float() + self.eps).rsqrt()
        return scale, self.bias.float() - self.running_mean.float() * scale

    def forward(self, x):
        return F.batch_norm(x, self.running_mean, self.running_var, self.weight, self.bias, training=False, eps=self.eps)

    def _load_from_state_dict(self, *a, **k):
        _PACK_EPOCH[0] += 1      # folded copies anywhere in the tree are stale
        return super()._load_from_state_dict(*a, **k)


def _fold(conv, bn):
    """(w (Cout, kh*kw*Cin) bf16 tap-major with the affine scale folded, shift (Cout) f32)."""
    scale, shift = bn.scale_shift()
    w = conv.weight.detach().float() * scale.view(-1, 1, 1, 1)
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous().to(torch.bfloat16), shift.contiguous()


class _Folded:
    """Per-module cache of folded weights, dropped when any state dict has been loaded since."""

    def _folded(self, key, make):
        st = self.__dict__.setdefault("_fold_cache", {})
        ent = st.get(key)
        if ent is None or ent[1] != _PACK_EPOCH[0] or ent[0][0].device != self._fold_device():
            ent = (make(), _PACK_EPOCH[0])
            st[key] = ent
        return ent[0]


def _run_conv(x, conv, wb, relu, residual=None):
    """x (N,1,H,W,Cin) bf16 -> (N,1,Ho,Wo,Cout) bf16."""
    w, b = wb
    kh, kw = conv.kernel_size
    return ops.conv3d(x, w, b, (1, kh, kw), (1,) + tuple(conv.stride), (0,) + tuple(conv.padding), (1,) + tuple(conv.dilation),
                      residual=residual, relu=relu)


class bottleneck_transformation(nn.Module, _Folded):
    """ResNet.py:254-301."""

    def __init__(self, inplanes, outplanes, innerplanes, stride=1, dilation=1, group=1, downsample=None, stride_1x1=False):
        super().__init__()
        if group != 1:
            raise NotImplementedError("grouped 3x3 convs (ResNeXt bodies) are outside TRACE's configs")
        str1x1, str3x3 = (stride, 1) if stride_1x1 else (1, stride)      # :262
        self.stride = stride
        self.conv1 = nn.Conv2d(inplanes, innerplanes, kernel_size=1, stride=str1x1, bias=False)
        self.bn1 = AffineChannel2d(innerplanes)
        self.conv2 = nn.Conv2d(innerplanes, innerplanes, kernel_size=3, stride=str3x3, bias=False, padding=1 * dilation,
                               dilation=dilation, groups=group)
        self.bn2 = AffineChannel2d(innerplanes)
        self.conv3 = nn.Conv2d(innerplanes, outplanes, kernel_size=1, stride=1, bias=False)
        self.bn3 = AffineChannel2d(outplanes)
        self.downsample = downsample
        self.relu = nn.ReLU(inplace=True)

    def _fold_device(self):
        return self.conv1.weight.device

    def forward(self, x):
        """Plain torch path (NCHW f32), the reference's arithmetic."""
        residual = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            residual = self.downsample(x)
        return self.relu(out + residual)

    def run(self, x):
        """x channels-last bf16 (N,1,H,W,C)."""
        w1, w2, w3 = (self._folded(i, lambda c=c, b=b: _fold(c, b)) for i, (c, b) in
                      enumerate(((self.conv1, self.bn1), (self.conv2, self.bn2), (self.conv3, self.bn3))))
        residual = x
        if self.downsample is not None:
            wd = self._folded("d", lambda: _fold(self.downsample[0], self.downsample[1]))
            residual = _run_conv(x, self.downsample[0], wd, relu=False)
        out = _run_conv(x, self.conv1, w1, relu=True)
        out = _run_conv(out, self.conv2, w2, relu=True)
        return _run_conv(out, self.conv3, w3, relu=True, residual=residual)      # relu(bn3(conv3) + residual)


def basic_bn_shortcut(inplanes, outplanes, stride):
    return nn.Sequential(nn.Conv2d(inplanes, outplanes, kernel_size=1, stride=stride, bias=False), AffineChannel2d(outplanes))


def basic_bn_stem():
    return nn.Sequential(OrderedDict([
        ("conv1", nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)),
        ("bn1", AffineChannel2d(64)),
        ("relu", nn.ReLU(inplace=True)),
        ("maxpool", nn.MaxPool2d(kernel_size=3, stride=2, padding=1))]))


def add_stage(inplanes, outplanes, innerplanes, nblocks, dilation=1, stride_init=2, stride_1x1=False):
    """ResNet.py:166-198."""
    blocks = []
    stride = stride_init
    for _ in range(nblocks):
        downsample = basic_bn_shortcut(inplanes, outplanes, stride) if (stride != 1 or inplanes != outplanes) else None
        blocks.append(bottleneck_transformation(inplanes, outplanes, innerplanes, stride, dilation=dilation, group=1,
                                                downsample=downsample, stride_1x1=stride_1x1))
        inplanes = outplanes
        stride = 1
    return nn.Sequential(*blocks), outplanes


class ResNet_convX_body(nn.Module, _Folded):
    """ResNet.py:49-123.  block_counts (3,4,6,3) / (3,4,23,3); dilation_num: res5 dilation (None: `res5_dilation`)."""
    CP = 4   # image channels padded 3 -> 4 (8-byte pixels), as the 3-D stem

    def __init__(self, block_counts, dilation_num=None, res5_dilation=2, stride_1x1=False, width=64):
        super().__init__()
        self.block_counts = block_counts
        self.convX = len(block_counts) + 1
        self.num_layers = (sum(block_counts) + 3 * (self.convX == 4)) * 3 + 2
        self.res1 = basic_bn_stem()
        dim_in = 64
        self.res2, dim_in = add_stage(dim_in, 256, width, block_counts[0], dilation=1, stride_init=1, stride_1x1=stride_1x1)
        self.res3, dim_in = add_stage(dim_in, 512, width * 2, block_counts[1], dilation=1, stride_init=2, stride_1x1=stride_1x1)
        self.res4, dim_in = add_stage(dim_in, 1024, width * 4, block_counts[2], dilation=1, stride_init=2, stride_1x1=stride_1x1)
        if len(block_counts) == 4:
            if dilation_num is None:
                dilation_num = res5_dilation
            stride_init = 2 if dilation_num == 1 else 1
            self.res5, dim_in = add_stage(dim_in, 2048, width * 8, block_counts[3], dilation_num, stride_init, stride_1x1=stride_1x1)
            self.spatial_scale = 1 / 32 * dilation_num
        else:
            self.spatial_scale = 1 / 16
        self.dim_out = dim_in
        for p in self.parameters():      # inference body: _init_modules freezes res1..FREEZE_AT and every affine layer
            p.requires_grad = False

    def _fold_device(self):
        return self.res1.conv1.weight.device

    def forward_torch(self, x):
        """The reference's forward (:120-123) on stock torch ops (NCHW f32)."""
        for i in range(self.convX):
            x = getattr(self, "res%d" % (i + 1))(x)
        return x

    def _stem_w(self):
        def make():
            scale, shift = self.res1.bn1.scale_shift()
            w = self.res1.conv1.weight.detach().float() * scale.view(-1, 1, 1, 1)          # (64,3,7,7)
            wp = torch.zeros((64, 1, 7, 8, self.CP), device=w.device)
            wp[:, 0, :, :7, :3] = w.permute(0, 2, 3, 1)
            return wp.reshape(64, -1).to(torch.bfloat16).contiguous(), shift.contiguous()
        return self._folded("stem", make)

    def forward_nhwc(self, x):
        """x (N,3,H,W) f32 -> channels-last bf16 (N, H/16, W/16, dim_out)."""
        if self.training:
            raise NotImplementedError("trace_amd.ResNet_convX_body implements the inference path; call .eval()")
        with torch.no_grad():
            N, C, H, W = x.shape
            w, b = self._stem_w()
            Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
            pw1 = max((Wo - 1) * 2 + 8 - (W + 3), 3)
            if (W + 3 + pw1) % 2:
                pw1 += 1
            xp = ops.clip_to_ndhwc(x.unsqueeze(2), False, self.CP, (0, 0, 3, 3, 3, pw1))       # (N,1,H+6,W+3+pw1,4) bf16
            y = ops.conv3d_stem(xp, w, b, 1, 7, 8, (1, 2, 2), (1, Ho, Wo), alg_k=7 * 7 * C)
            y = ops.maxpool3d(y, (1, 3, 3), (1, 2, 2), (0, 1, 1))
            for i in range(2, self.convX + 1):
                for blk in getattr(self, "res%d" % i):
                    y = blk.run(y)
        return y[:, 0]

    def forward(self, x):
        """Reference signature (:120-123): (N,3,H,W) f32 -> (N, dim_out, H/16, W/16) f32."""
        if not x.is_cuda:
            return self.forward_torch(x)
        return self.forward_nhwc(x).permute(0, 3, 1, 2).float()


def ResNet50_conv5_body_dilation2():
    """ResNet.py:24-25 - TRACE's Prd_RCNN2D (MODEL.REL_CONV_BODY)."""
    return ResNet_convX_body((3, 4, 6, 3), dilation_num=2)


def ResNet101_conv5_body(res5_dilation=2):
    """ResNet.py:31-32 with RESNETS.RES5_DILATION 2 of the VidVRD / AG configs - TRACE's Conv_Body."""
    return ResNet_convX_body((3, 4, 23, 3), res5_dilation=res5_dilation)


def ResNet101_conv5_body_dilation2():
    return ResNet_convX_body((3, 4, 23, 3), dilation_num=2)


def ResNet50_conv5_body(res5_dilation=2):
    return ResNet_convX_body((3, 4, 6, 3), res5_dilation=res5_dilation)
