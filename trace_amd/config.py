"""The handful of `cfg` keys the hot path reads (lib/core/config.py:1144-1177, :733, :244 and
configs/vidvrd/vidvrd_res101xi3d50_gt_boxes_dc5_2d_new.yaml), as a small dataclass instead of
the reference's global mutable AttrDict."""
from dataclasses import dataclass


@dataclass
class TraceConfig:
    NUM_CLASSES: int = 36            # tools/test_net_rel.py:156-160 (vidvrd: 35 + background)
    NUM_PRD_CLASSES: int = 132
    MLP_HEAD_DIM: int = 1024         # FAST_RCNN.MLP_HEAD_DIM
    ROI_XFORM_RESOLUTION: int = 7
    ROI_XFORM_SAMPLING_RATIO: int = 2
    RESIZE_HIDDEN_DIM: int = 4
    MSP_GROUP: int = 4
    MSP_MAX_TREE_HEIGHT: int = 7
    CLUSTER_NUM_PARTITION: int = 2
    NO_FC7_RELU: bool = True
    MULTI_RELATION: bool = True
    USE_FREQ_BIAS: bool = True
    USE_OVLP_FILTER: bool = False
    I3D_DC5: bool = True
    HALF_NUMBER_OF_FRAMES: int = 8
    CONV_BODY_DIM: int = 2048        # ResNet101_conv5_body / ResNet50_conv5_body_dilation2 dim_out
    CONV_BODY_SCALE: float = 1.0 / 16
    REL_SPATIAL_SCALE: float = 1.0 / 16
    OBJ_EMBED_DIM: int = 200
    LOW_FEAT_DIM: int = 256
    DETECTIONS_PER_IM: int = 100
    TEST_NMS: float = 0.5
    TEST_SCORE_THRESH: float = 0.05   # lib/core/config.py TEST.SCORE_THRESH
    FRAMES_INTERVAL: int = 4          # configs/vidvrd/...gt_boxes...yaml:69-70 (HALF_NUMBER_OF_FRAMES 8, interval 4)
    TEMPORAL_SCALES: int = 224        # short side of the clip frames (test_engine_rel.py:468)
    TEST_SCALE: int = 800             # key-frame short side / cap (TEST.SCALE, TEST.MAX_SIZE)
    TEST_MAX_SIZE: int = 1333
    dataset: str = "vidvrd"


def vidvrd_config():
    return TraceConfig()


def ag_config():
    """Action Genome dims (tools/test_net_rel.py:166-170; configs/ag/res101xi3d50_dc5_2d.yaml)."""
    return TraceConfig(NUM_CLASSES=37, NUM_PRD_CLASSES=26, RESIZE_HIDDEN_DIM=1, USE_OVLP_FILTER=True, dataset="ag")


# The reference's modules read a global `cfg` (lib/core/config.py:40); constructors here take `cfg=` explicitly and
# fall back to this process-wide default, which a caller may replace (`set_cfg`) before building modules.
_DEFAULT = [None]


def get_cfg():
    if _DEFAULT[0] is None:
        _DEFAULT[0] = TraceConfig()
    return _DEFAULT[0]


def set_cfg(cfg):
    _DEFAULT[0] = cfg
    return cfg
