"""Synthetic workload of SURVEY.md 8(d) (seeded): clips, key-frame / 2-D feature maps, boxes and a
random-initialised model with a synthetic frequency-bias table.  No dataset or checkpoint is
available offline, so bench.py, smoke() and the tests all draw their inputs from here."""
import numpy as np
import torch


def synth_boxes(rng, n, k=2.0, size=224.0):
    """x1,y1 ~ U(0,160), w,h ~ U(16,64k) clipped to size-1."""
    x1 = rng.uniform(0, 160, n)
    y1 = rng.uniform(0, 160, n)
    w = rng.uniform(16, 64 * k, n)
    h = rng.uniform(16, 64 * k, n)
    return np.stack([x1, y1, np.minimum(x1 + w, size - 1), np.minimum(y1 + h, size - 1)], 1).astype(np.float32)


def make_inputs(B, T, n_obj, hw=224, seed=1, device="cpu"):
    """all_frames (B,T,3,hw,hw) ~ N(0,1)*50; blob_conv / feat_map (B,2048,hw/16,hw/16) ~ N(0,1);
    det_rois: list of B (n,5) arrays."""
    g = torch.Generator().manual_seed(seed)
    rng = np.random.default_rng(seed)
    frames = torch.randn((B, T, 3, hw, hw), generator=g) * 50
    fm = hw // 16
    blob = torch.randn((B, 2048, fm, fm), generator=g)
    feat = torch.randn((B, 2048, fm, fm), generator=g)
    rois = [np.concatenate([np.zeros((n_obj, 1), np.float32), synth_boxes(rng, n_obj, 2.0, float(hw))], 1)
            for _ in range(B)]
    return frames.to(device), blob.to(device), feat.to(device), rois


def randomize_model(model, seed=1, calib_clip=None, bn_calibrate=None):
    """Seeded BN affine params + freq-bias table ~ log U(1e-3,1).  `bn_calibrate`
    (test/bench infrastructure) may replace the backbone state dict with calibrated BN statistics."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():   # in-place writes that bump the version counters the packed-weight caches watch
        _randomize(model, g)
    return model


def _randomize(model, g):
    for name, mod in model.named_modules():
        if isinstance(mod, torch.nn.BatchNorm3d):
            lo, hi = (0.2, 0.5) if (name.endswith("conv3.bn") or "conv_out" in name) else (0.7, 1.3)
            mod.weight.copy_(torch.rand(mod.num_features, generator=g) * (hi - lo) + lo)
            mod.bias.copy_(torch.randn(mod.num_features, generator=g) * 0.1)
            mod.running_mean.copy_(torch.randn(mod.num_features, generator=g) * 0.1)
            mod.running_var.copy_(torch.rand(mod.num_features, generator=g) * 0.5 + 0.75)
    if hasattr(model, "freq_bias"):
        tab = model.freq_bias.rel_baseline.weight
        tab.copy_(torch.log(torch.rand(tab.shape, generator=g) * (1 - 1e-3) + 1e-3))
    return model


def fill_resnet2d_state(module, seed=1):
    """Seeded weights for a 2-D ResNet body (trace_amd.resnet2d or the reference's ResNet_convX_body: same key names),
    one generator per state-dict key so that any implementation holding the same keys gets the same tensors:
    conv ~ N(0, 2/fan_in); affine weight U(.7,1.3) (bn3 / shortcut: U(.2,.5), keeps the residual stream bounded),
    bias, running_mean ~ N(0,.1), running_var U(.75,1.25)."""
    import zlib
    sd = module.state_dict()
    out = {}
    for k, v in sd.items():
        g = torch.Generator().manual_seed((zlib.crc32(k.encode()) + 7919 * seed) & 0x7fffffff)
        leaf = k.rsplit(".", 1)[-1]
        if v.dim() == 4:
            t = torch.randn(v.shape, generator=g) * (2.0 / (v.shape[1] * v.shape[2] * v.shape[3])) ** 0.5
        elif leaf == "weight":
            lo, hi = (0.2, 0.5) if (".bn3." in k or ".downsample." in k) else (0.7, 1.3)
            t = torch.rand(v.shape, generator=g) * (hi - lo) + lo
        elif leaf == "running_var":
            t = torch.rand(v.shape, generator=g) * 0.5 + 0.75
        else:
            t = torch.randn(v.shape, generator=g) * 0.1
        out[k] = t.to(v.dtype)
    module.load_state_dict(out)
    return module
