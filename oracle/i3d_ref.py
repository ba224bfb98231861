"""TEST INFRASTRUCTURE ONLY — fp32 PyTorch (CPU) restatement of the I3D ResNet-50
clip backbone, driven by a plain state_dict with the reference's key names.

Follows lib/i3d/resnet3d.py: stem :718-737, forward :785-802, Bottleneck3d
:155-318, make_res_layer :486-598, with the DC5 overrides of
lib/modeling_rel/model_builder_rel.py:166-183 and the backbone dict of
lib/i3d/i3d_nl_dot_product_r50_32x2x1_100e_kinetics400_rgb.py:4-20.

mmcv (ConvModule, NonLocal3d) is a third-party dependency absent from
/root/reference and from this image (version unpinned, README.md:24; 1.x implied
by `mmcv.runner`).  Its published algorithm is restated here: ConvModule =
conv -> BN (eval statistics) -> ReLU; NonLocal3d(dot_product, sub_sample):
theta/phi/g 1x1x1 convs with bias, phi and g max-pooled (1,2,2),
pairwise = theta^T phi / N_kv, y = pairwise g, out = x + BN(conv_out(y)).
PARITY UNPINNED for these blocks: no reference test or fixture exercises them.
"""
import torch
import torch.nn.functional as F

STAGE_BLOCKS = (3, 4, 6, 3)
INFLATE = ((1, 1, 1), (1, 0, 1, 0), (1, 0, 1, 0, 1, 0), (0, 1, 0))
NON_LOCAL = ((0, 0, 0), (0, 1, 0, 1), (0, 1, 0, 1, 0, 1), (0, 0, 0))
DILATIONS = (1, 1, 1, 2)
SPATIAL_STRIDES = (1, 2, 2, 1)


_CALIBRATE = [False]
_EMULATE_BF16 = [False]


def _rb(x):
    return x.bfloat16().float()


def _conv_module(sd, prefix, x, stride=1, padding=0, dilation=1, relu=True):
    w = sd[prefix + ".conv.weight"]
    b = sd.get(prefix + ".conv.bias")
    if _EMULATE_BF16[0]:
        # same fp32 algorithm with every tensor that a bf16 pipeline stores rounded to bf16:
        # BN folded into the weights (then rounded), activations rounded after the epilogue.
        if prefix + ".bn.weight" in sd:
            scale = sd[prefix + ".bn.weight"] / torch.sqrt(sd[prefix + ".bn.running_var"] + 1e-5)
            shift = sd[prefix + ".bn.bias"] - sd[prefix + ".bn.running_mean"] * scale
            if b is not None:
                shift = shift + b * scale
            w = w * scale.view(-1, 1, 1, 1, 1)
        else:
            shift = b
        x = F.conv3d(x, _rb(w), shift, stride=stride, padding=padding, dilation=dilation)
        return x  # caller rounds after residual/ReLU
    x = F.conv3d(x, w, b, stride=stride, padding=padding, dilation=dilation)
    if prefix + ".bn.weight" in sd:
        if _CALIBRATE[0]:  # synthetic-weight helper: set the running statistics to this batch's
            sd[prefix + ".bn.running_mean"] = x.mean(dim=(0, 2, 3, 4))
            v = x.var(dim=(0, 2, 3, 4), unbiased=False)
            sd[prefix + ".bn.running_var"] = torch.clamp(v, min=0.1 * float(v.mean()) + 1e-6)
        x = F.batch_norm(x, sd[prefix + ".bn.running_mean"], sd[prefix + ".bn.running_var"],
                         sd[prefix + ".bn.weight"], sd[prefix + ".bn.bias"], False, 0.0, 1e-5)
    return F.relu(x) if relu else x


def _cm(sd, prefix, x, stride=1, padding=0, dilation=1, relu=True):
    """_conv_module + (in emulation mode) ReLU and bf16 rounding of the stored activation."""
    y = _conv_module(sd, prefix, x, stride, padding, dilation, relu)
    if _EMULATE_BF16[0]:
        y = _rb(F.relu(y) if relu else y)
    return y


def _non_local(sd, prefix, x):
    n = x.size(0)
    g_x = F.max_pool3d(_cm(sd, prefix + ".g.0", x, relu=False), (1, 2, 2))
    inter = g_x.size(1)
    g_x = g_x.view(n, inter, -1).permute(0, 2, 1)
    theta_x = _cm(sd, prefix + ".theta", x, relu=False).view(n, inter, -1).permute(0, 2, 1)
    phi_x = F.max_pool3d(_cm(sd, prefix + ".phi.0", x, relu=False), (1, 2, 2)).view(n, inter, -1)
    pairwise = torch.matmul(theta_x, phi_x)
    pairwise = pairwise / pairwise.shape[-1]
    if _EMULATE_BF16[0]:
        pairwise = _rb(pairwise)
    y = torch.matmul(pairwise, g_x).permute(0, 2, 1).contiguous().reshape(n, inter, *x.size()[2:])
    if _EMULATE_BF16[0]:
        return _rb(x + _conv_module(sd, prefix + ".conv_out", _rb(y), relu=False))
    return x + _conv_module(sd, prefix + ".conv_out", y, relu=False)


def _bottleneck(sd, prefix, x, spatial_stride, dilation, inflate, non_local):
    # resnet3d.py:222-248 (style='pytorch', inflate_style='3x1x1'), :292-318
    if inflate:
        out = _cm(sd, prefix + ".conv1", x, padding=(1, 0, 0))
    else:
        out = _cm(sd, prefix + ".conv1", x)
    out = _cm(sd, prefix + ".conv2", out, stride=(1, spatial_stride, spatial_stride),
              padding=(0, dilation, dilation), dilation=(1, dilation, dilation))
    out = _conv_module(sd, prefix + ".conv3", out, relu=False)  # rounded after the fused add + ReLU
    identity = x
    if prefix + ".downsample.conv.weight" in sd:
        identity = _cm(sd, prefix + ".downsample", x, stride=(1, spatial_stride, spatial_stride), relu=False)
    out = F.relu(out + identity)
    if _EMULATE_BF16[0]:
        out = _rb(out)
    if non_local:
        out = _non_local(sd, prefix + ".non_local_block", out)
    return out


def wiring():
    """Every Conv3d of the network as this file runs it: name -> (kernel, stride, padding, dilation), and the blocks
    that carry a non-local block.  tests/test_oracle_goldens_r2.py compares it with the constructor arguments recorded
    from the reference's own lib/i3d/resnet3d.py (tests/golden/py_i3d_wiring.npz)."""
    convs = {"conv1": ((5, 7, 7), (2, 2, 2), (2, 3, 3), (1, 1, 1))}
    nl = []
    for i, nb in enumerate(STAGE_BLOCKS):
        for b in range(nb):
            p = "layer%d.%d" % (i + 1, b)
            ss = SPATIAL_STRIDES[i] if b == 0 else 1
            d = DILATIONS[i]
            if INFLATE[i][b] == 1:
                convs[p + ".conv1"] = ((3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1))
            else:
                convs[p + ".conv1"] = ((1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1))
            convs[p + ".conv2"] = ((1, 3, 3), (1, ss, ss), (0, d, d), (1, d, d))
            convs[p + ".conv3"] = ((1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1))
            if b == 0:
                convs[p + ".downsample"] = ((1, 1, 1), (1, ss, ss), (0, 0, 0), (1, 1, 1))
            if NON_LOCAL[i][b] == 1:
                nl.append(p + ".non_local_block")
    pools = {"maxpool": ((1, 3, 3), (2, 2, 2), (0, 1, 1)), "pool2": ((2, 1, 1), (2, 1, 1), (0, 0, 0))}
    return convs, nl, pools


def i3d_forward(sd, x, return_stages=False, emulate_bf16=False):
    """x (N,3,T,H,W) f32 -> (N,2048,T',H/16,W/16) f32.

    emulate_bf16=True runs the SAME fp32 algorithm but rounds to bf16 every tensor a bf16
    pipeline stores (folded weights, clip, each layer's activation, the non-local pairwise
    matrix): the precision-matched checker for the bf16 MFMA path."""
    sd = {k: v.float().cpu() for k, v in sd.items()}
    _EMULATE_BF16[0] = bool(emulate_bf16)
    try:
        return _forward(sd, _rb(x) if emulate_bf16 else x, return_stages)
    finally:
        _EMULATE_BF16[0] = False


def _forward(sd, x, return_stages):
    stages = []
    x = _cm(sd, "conv1", x, stride=(2, 2, 2), padding=(2, 3, 3))
    x = F.max_pool3d(x, (1, 3, 3), (2, 2, 2), (0, 1, 1))
    stages.append(x)
    for i, nb in enumerate(STAGE_BLOCKS):
        for b in range(nb):
            x = _bottleneck(sd, "layer%d.%d" % (i + 1, b), x, SPATIAL_STRIDES[i] if b == 0 else 1, DILATIONS[i],
                            INFLATE[i][b] == 1, NON_LOCAL[i][b] == 1)
        if i == 0:
            x = F.max_pool3d(x, (2, 1, 1), (2, 1, 1))
        stages.append(x)
    return (x, stages) if return_stages else x


def calibrate_bn(sd, x):
    """Synthetic-weight helper (no reference counterpart): returns a copy of `sd` whose BatchNorm
    running statistics are the batch statistics of clip `x`, so that randomly initialised weights
    give O(1) activations as a trained checkpoint would."""
    sd = {k: v.detach().float().cpu().clone() for k, v in sd.items()}
    _CALIBRATE[0] = True
    try:
        x = _conv_module(sd, "conv1", x, stride=(2, 2, 2), padding=(2, 3, 3))
        x = F.max_pool3d(x, (1, 3, 3), (2, 2, 2), (0, 1, 1))
        for i, nb in enumerate(STAGE_BLOCKS):
            for b in range(nb):
                x = _bottleneck(sd, "layer%d.%d" % (i + 1, b), x, SPATIAL_STRIDES[i] if b == 0 else 1, DILATIONS[i],
                                INFLATE[i][b] == 1, NON_LOCAL[i][b] == 1)
            if i == 0:
                x = F.max_pool3d(x, (2, 1, 1), (2, 1, 1))
    finally:
        _CALIBRATE[0] = False
    return sd


def i3d_stage(sd, i, x):
    """fp32 forward of residual stage i (0-based: layer{i+1}, plus pool2 after stage 0)."""
    sd = {k: v.float().cpu() for k, v in sd.items()}
    for b in range(STAGE_BLOCKS[i]):
        x = _bottleneck(sd, "layer%d.%d" % (i + 1, b), x, SPATIAL_STRIDES[i] if b == 0 else 1, DILATIONS[i],
                        INFLATE[i][b] == 1, NON_LOCAL[i][b] == 1)
    if i == 0:
        x = F.max_pool3d(x, (2, 1, 1), (2, 1, 1))
    return x
