"""trace_amd.io: checkpoint key maps / load_ckpt_rel against the reference's own functions
(tests/golden/py_eval.npz ckpt_* / tiny_*, tools/make_goldens_r2c.py); frame window and blob preparation (the
reference's path needs OpenCV: parity unpinned, checked through its defining properties)."""
import json
import os

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = np.load(os.path.join(ROOT, "tests", "golden", "py_eval.npz"))


def jl(a):
    return json.loads(bytes(a).decode())


def test_key_maps_equal_reference_on_every_model_key():
    from trace_amd import io
    keys = jl(G["ckpt_keys"])
    assert [io.make_res101_map(k) for k in keys] == jl(G["ckpt_res101"])
    assert [io.make_res50_fpn_map(k) for k in keys] == jl(G["ckpt_res50fpn"])
    assert [io.make_backbone_map(k) for k in keys] == jl(G["ckpt_backbone"])
    c2a, idx = io.class_coco2ag()
    assert np.array_equal(c2a, G["c2a"]) and np.array_equal(idx, G["c2a_idx"])
    # the keys the generator used ARE this model's keys (plus the FPN-style extras at the end)
    from trace_amd.model_builder_rel import Generalized_RCNN
    from trace_amd.rpn import single_scale_rpn_outputs
    m = Generalized_RCNN()
    m.RPN = single_scale_rpn_outputs(2048, 1 / 16)
    mine = list(m.state_dict().keys())
    assert keys[:len(mine)] == mine


def test_load_ckpt_rel_equals_reference():
    from trace_amd import io

    class Tiny(nn.Module):
        def __init__(self):
            super().__init__()
            self.Box_Outs = nn.Module()
            self.Box_Outs.cls_score = nn.Linear(8, 37)
            self.Box_Head = nn.Module()
            self.Box_Head.fc1 = nn.Linear(8, 4)
            self.other = nn.Linear(3, 3)
    t = Tiny()
    for p in t.parameters():
        p.data.zero_()
    ck = {k[len("tiny_ck_"):]: G[k] for k in G.files if k.startswith("tiny_ck_")}
    loaded = io.load_ckpt_rel(t, ck)
    for k, v in t.state_dict().items():
        assert np.array_equal(v.numpy(), G["tiny_sd_" + k]), k
    assert "Box_Head.fc1.bias" not in loaded and "other.weight" in loaded and "Box_Head.fc1.weight" in loaded
    assert "Box_Outs.cls_score.weight" not in loaded        # 81 COCO rows vs 37: the reference's re-indexing writes to a copy
    assert io.trans_prd({"module.Prd_RCNN.conv1.conv.weight": np.zeros(2), "x": np.ones(1)}).keys() == {"conv1.conv.weight"}


def test_frame_window():
    from trace_amd.io import frame_window
    # VidVRD config: HALF_NUMBER_OF_FRAMES 8, FRAMES_INTERVAL 4 -> 17 frames, key frame in the middle
    ids, bf = frame_window(100, 8, 4)
    assert ids == list(range(68, 133, 4)) and bf == 8 and ids[bf] == 100
    # near the start of the video: fewer frames before the key frame
    ids, bf = frame_window(9, 8, 4, im_id_st=1)
    assert ids == [1, 5, 9] + list(range(13, 42, 4)) and bf == 2 and ids[bf] == 9
    # frames missing after the key frame: the walk stops at the first gap
    ids, bf = frame_window(100, 8, 4, exists=lambda j: j <= 112)
    assert ids == list(range(68, 113, 4)) and bf == 8
    # the reference's defaults (3, 3)
    ids, bf = frame_window(30, 3, 3)
    assert ids == [21, 24, 27, 30, 33, 36, 39] and bf == 3


def test_clip_from_frames():
    from trace_amd.io import PIXEL_MEANS, clip_from_frames, get_target_scale
    assert get_target_scale(240, 320, 320, 1000) == 320 / 240 and get_target_scale(500, 2000, 800, 1333) == 1333 / 2000
    rng = np.random.default_rng(0)
    fr = rng.integers(0, 256, (5, 224, 224, 3), dtype=np.uint8)
    blob, s = clip_from_frames(fr, 224)
    assert s == 1.0 and blob.shape == (1, 5, 3, 224, 224) and blob.dtype == torch.float32
    want = fr.astype(np.float32) - np.asarray(PIXEL_MEANS, np.float32)
    assert np.array_equal(blob[0].permute(0, 2, 3, 1).numpy(), want)
    u8, s = clip_from_frames(list(fr), 224, keep_uint8=True)
    assert u8.dtype == torch.uint8 and u8.shape == (1, 5, 224, 224, 3) and np.array_equal(u8[0].numpy(), fr)
    # resize: output size = round(size * scale); a constant image stays constant; means are subtracted before resizing
    fr2 = np.full((2, 240, 320, 3), 200, np.uint8)
    blob, s = clip_from_frames(fr2, 320)
    assert s == 320 / 240 and blob.shape == (1, 2, 3, 320, 427)
    np.testing.assert_allclose(blob[0, :, 1].numpy(), 200 - PIXEL_MEANS[1], atol=1e-4)
