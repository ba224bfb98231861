"""2-D conv bodies on the GPU (implicit-GEMM convs, AffineChannel2d folded) against the fp32 oracle."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-12))


@pytest.mark.parametrize("ctor,counts,shape", [
    ("ResNet50_conv5_body_dilation2", (3, 4, 6, 3), (2, 3, 224, 224)),     # Prd_RCNN2D on the relation frames
    ("ResNet50_conv5_body_dilation2", (3, 4, 6, 3), (1, 3, 150, 203)),     # odd sizes: every halo / stride remainder
    ("ResNet101_conv5_body", (3, 4, 23, 3), (1, 3, 256, 352)),             # Conv_Body (a reduced key frame)
])
def test_body_matches_fp32_oracle(ctor, counts, shape):
    from oracle.resnet2d_ref import body_forward
    from trace_amd import resnet2d
    from trace_amd.synthetic import fill_resnet2d_state
    net = fill_resnet2d_state(getattr(resnet2d, ctor)().eval(), seed=5)
    g = torch.Generator().manual_seed(6)
    img = torch.randn(shape, generator=g) * 50
    ref = body_forward(net.state_dict(), img, counts)
    net.to(DEV)
    got = net(img.to(DEV))
    assert got.shape == ref.shape and got.dtype == torch.float32
    e = rel_l2(got.cpu(), ref)
    assert e < 2e-2, (ctor, shape, e)                                       # north_star: features within 2e-2 (bf16) of fp32
    nhwc = net.forward_nhwc(img.to(DEV))
    assert nhwc.dtype == torch.bfloat16 and torch.equal(nhwc.permute(0, 3, 1, 2).float(), got)
    # reloading weights through load_state_dict refolds the cached bf16 copies
    fill_resnet2d_state(net, seed=8)
    ref2 = body_forward({k: v.cpu() for k, v in net.state_dict().items()}, img, counts)
    assert rel_l2(net(img.to(DEV)).cpu(), ref2) < 2e-2


def test_model_forward_runs_its_own_conv_bodies():
    """Generalized_RCNN.forward (reference signature) computes blob_conv / feat_map with its default Conv_Body /
    Prd_RCNN2D (model_builder_rel.py:702, :1006-1007) and gives what forward_clips gives on those maps."""
    import numpy as np
    from trace_amd import resnet2d
    from trace_amd.model_builder_rel import Generalized_RCNN
    from trace_amd.synthetic import fill_resnet2d_state, make_inputs, randomize_model
    m = Generalized_RCNN()
    assert isinstance(m.Conv_Body, resnet2d.ResNet_convX_body) and m.Conv_Body.block_counts == (3, 4, 23, 3)
    assert m.Prd_RCNN2D.block_counts == (3, 4, 6, 3) and m.Conv_Body.spatial_scale == m.Prd_RCNN2D.spatial_scale == 1 / 16
    sd_keys = set(m.state_dict())
    assert "Conv_Body.res5.2.bn3.running_var" in sd_keys and "Prd_RCNN2D.res1.conv1.weight" in sd_keys
    randomize_model(m, 1)
    fill_resnet2d_state(m.Conv_Body, 2)
    fill_resnet2d_state(m.Prd_RCNN2D, 3)
    m.eval().to(DEV)
    frames, _, _, rois = make_inputs(1, 8, 6, 224, seed=2)
    frames = frames.to(DEV)
    data = frames[:, 3].contiguous()
    blob, feat = m.Conv_Body(data), m.Prd_RCNN2D(frames[:, 4])
    assert blob.shape == feat.shape == (1, 2048, 14, 14)
    want = m.forward_clips(frames, blob, feat, rois, topk=50, prd_k=1)[0]
    want = {k: want[k].clone() for k in ("prd_scores",)}
    ret = m(data, frames, torch.tensor([[224.0, 224.0, 1.0]]),
            pre_processed_frames_rpn_ret=dict(rois=np.asarray(rois[0], np.float32).copy()), dataset_name="vidvrd_val", bf_cur_len=4)
    assert torch.equal(ret["prd_scores"], want["prd_scores"]) and ret["prd_scores"].shape[0] == 6 * 5
    slim = Generalized_RCNN(conv_bodies=False)
    assert slim.Conv_Body is None and not any(k.startswith("Conv_Body") for k in slim.state_dict())
