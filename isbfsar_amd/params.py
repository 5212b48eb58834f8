"""Configuration classes mirroring the reference's ``utils/params.py`` (same field names and
defaults for the fields the hot path consumes), plus the fields the HIP build adds.

    MainConfig            utils/params.py:14-22
    MetrabsTRTConfig      utils/params.py:25-37   -> MetrabsHIPConfig (engines become one weight blob)
    RealSenseIntrinsics   utils/params.py:40-47
    TRXConfig             utils/params.py:50-95
"""
import os

input_type = "skeleton"            # utils/params.py:4
skeleton_type = "smpl+head_30"     # utils/params.py:5
seq_len = 16                       # utils/params.py:8 (skeleton mode)

_ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")


class MainConfig(object):
    def __init__(self):
        self.input_type = input_type
        self.cam = "realsense"
        self.cam_width = 640
        self.cam_height = 480
        self.window_size = seq_len
        self.skeleton_scale = 2200.
        self.acquisition_time = 3


class MetrabsHIPConfig(object):
    """Replaces MetrabsTRTConfig: the four TensorRT engines (yolo / image_transformation /
    bbone / heads, utils/params.py:27-30) become kernels inside libisbfsar_hip.so; the backbone
    and head weights arrive as one ISBW blob (``weights`` / ``weights_path``)."""

    def __init__(self):
        self.weights = None                    # mapping name -> ndarray, or bytes (ISBW)
        self.weights_path = None               # path to an .isbw blob
        self.weights_seed = 0                  # used when neither is given: deterministic synthetic weights
        self.yolo_weights = None               # YOLOv4 detector (the reference's yolo_engine_path, params.py:27): mapping / ISBW
        self.yolo_weights_path = None          # bytes / path to an .isbw blob (isbfsar_amd/yolov4.py); none of them set: no
        self.yolo_synthetic = False            # built-in detector (boxes come from bbox_provider / fixed_bbox / the whole frame)
        self.expand_joints_path = os.path.join(_ASSETS, "32_to_122.npy")
        self.skeleton_types_path = os.path.join(_ASSETS, "skeleton_types.json")
        self.skeleton = skeleton_type
        self.yolo_thresh = 0.3
        self.nms_thresh = 0.7
        self.num_aug = 0
        self.just_box = input_type == "rgb"
        self.device = 0
        self.max_batch = 64


MetrabsTRTConfig = MetrabsHIPConfig   # name kept so `from utils.params import MetrabsTRTConfig` ports 1:1


class RealSenseIntrinsics(object):
    def __init__(self):
        self.fx = 384.025146484375
        self.fy = 384.025146484375
        self.ppx = 319.09661865234375
        self.ppy = 237.75723266601562
        self.width = 640
        self.height = 480


class TRXConfig(object):
    def __init__(self):
        self.model = "DISC"
        self.input_type = input_type
        self.way = 5
        self.shot = 1
        self.device = 'cuda'
        self.skeleton_type = skeleton_type
        self.n_joints = 30
        self.trans_linear_in_dim = 256
        self.trans_linear_out_dim = 128
        self.query_per_class = 1
        self.trans_dropout = 0.
        self.temp_set = [2]
        self.final_ckpt_path = "modules/ar/modules/raws/DISC.pth"
        self.seq_len = seq_len
        # --- additions of the HIP build ---
        self.weights = None          # mapping name -> ndarray or ISBW bytes; overrides final_ckpt_path
        self.device_index = 0        # HIP device ordinal
        self.precision = "default"   # tuple-attention operands: "default" (= the library's: "f16") | "f16" | "bf16" | "bf16x3"
        self.max_batch = 1024
