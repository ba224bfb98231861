"""TEST INFRASTRUCTURE - CPU restatement (fp32, stock torch functional ops) of the reference's 2-D ResNet bodies.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.

Follows Detectron_pytorch/lib/modeling/ResNet.py over a state dict with the reference's key names:
`basic_bn_stem` (:232-238), `add_stage` / `add_residual_block` (:166-198: the first block of a stage carries the stride
and a `basic_bn_shortcut` (:205-213) whenever stride != 1 or the width changes), `bottleneck_transformation`
(:254-301, stride on the 3x3 with RESNETS.STRIDE_1X1 False, `padding = dilation` on the 3x3), `ResNet_convX_body`
(:49-123: res5 dilated and un-strided when RES5_DILATION / dilation_num is 2), frozen BN `AffineChannel2d`
(nn/modules/affine.py:28-37).
Pinned by tests/golden/py_resnet2d.npz (outputs of the reference's own ResNet.py, tools/make_goldens_r2b.py).
"""
import torch
import torch.nn.functional as F


def affine(x, sd, p, eps=1e-5):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        training=False, eps=eps)


def bottleneck(x, sd, p, stride, dilation, stride_1x1=False):
    s1, s3 = (stride, 1) if stride_1x1 else (1, stride)
    out = F.relu(affine(F.conv2d(x, sd[p + ".conv1.weight"], stride=s1), sd, p + ".bn1"))
    out = F.relu(affine(F.conv2d(out, sd[p + ".conv2.weight"], stride=s3, padding=dilation, dilation=dilation), sd, p + ".bn2"))
    out = affine(F.conv2d(out, sd[p + ".conv3.weight"]), sd, p + ".bn3")
    res = x
    if p + ".downsample.0.weight" in sd:
        res = affine(F.conv2d(x, sd[p + ".downsample.0.weight"], stride=stride), sd, p + ".downsample.1")
    return F.relu(out + res)


def body_forward(sd, x, block_counts, res5_dilation=2, stride_1x1=False, stages_out=None):
    """x (N,3,H,W) f32 -> (N, 2048, H/16, W/16) (res5_dilation 2) f32."""
    sd = {k: v.float() for k, v in sd.items()}
    x = F.relu(affine(F.conv2d(x, sd["res1.conv1.weight"], stride=2, padding=3), sd, "res1.bn1"))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    if stages_out is not None:
        stages_out.append(x)
    for si, n in enumerate(block_counts):
        stage = "res%d" % (si + 2)
        dil = res5_dilation if si == 3 else 1
        stride_init = 1 if si == 0 else (2 if (si < 3 or dil == 1) else 1)
        for b in range(n):
            x = bottleneck(x, sd, "%s.%d" % (stage, b), stride_init if b == 0 else 1, dil, stride_1x1)
        if stages_out is not None:
            stages_out.append(x)
    return x
