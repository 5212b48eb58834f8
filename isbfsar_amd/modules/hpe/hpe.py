"""Drop-in for the reference's ``modules/hpe/hpe.py::HumanPoseEstimator`` (lines 15-173) on top
of libisbfsar_hip.so.  Same constructor and ``estimate(frame)`` contract, so the reference's
worker loop (``main.py:336-342``: ``x = module(*configurations); y = x.estimate(input_queue.get())``)
and the per-frame consumer (``main.py:77-105``) run unmodified:

    estimate(frame uint8[480,640,3] BGR) -> None | {"pose": float64[n,3], "edges": [...], "bbox": (x1,x2,y1,y2)}
    just_box mode                         -> {"bbox": (x1,y1,x2,y2)}            (hpe.py:82-83)

The person box comes, in this order of precedence, from: ``detector(frame) -> (boxes, confs)`` (any callable returning
the YOLO export tensors); the built-in YOLOv4 detector (``isb_det_*``: pre-processing hpe.py:51-56 + network hpe.py:59,
used when ``model_config.yolo_weights`` / ``yolo_weights_path`` / ``yolo_synthetic`` is set); ``bbox_provider(frame) ->
(x1,x2,y1,y2) | None``; ``model_config.fixed_bbox``; the whole frame. Detector outputs are post-processed on the GPU
(hpe.py:60-79). Everything runs on the GPU; there is no CPU path.
"""
from __future__ import annotations

import numpy as np

from ... import effnetv2
from ...hpe_engine import HpeEngine, load_joint_assets
from ...weights import unpack_blob


class HumanPoseEstimator:
    def __init__(self, model_config, cam_config, just_box=None, bbox_provider=None, detector=None):
        if just_box is None:
            self.just_box = model_config.just_box
        else:
            self.just_box = just_box
        self.yolo_thresh = model_config.yolo_thresh
        self.nms_thresh = model_config.nms_thresh
        self.num_aug = model_config.num_aug
        self.n_test = 1 if self.num_aug < 1 else self.num_aug     # hpe.py:25

        # Intrinsics and K matrix of RealSense (hpe.py:28-33)
        self.K = np.zeros((3, 3), np.float32)
        self.K[0][0] = cam_config.fx
        self.K[0][2] = cam_config.ppx
        self.K[1][1] = cam_config.fy
        self.K[1][2] = cam_config.ppy
        self.K[2][2] = 1

        self.skeleton = model_config.skeleton
        self.expand_joints, indices, self.edges = load_joint_assets(
            model_config.expand_joints_path, model_config.skeleton_types_path, self.skeleton)
        self.bbox_provider = bbox_provider
        self.detector = detector          # callable(frame) -> (boxes [1,4032,1,4], confs [1,4032,80]) like Runner(yolo)
        self.fixed_bbox = getattr(model_config, "fixed_bbox", None)

        # the reference's Runner(yolo_engine_path) (hpe.py:42): the built-in YOLOv4 network, when weights are configured
        self.det = None
        yw = getattr(model_config, "yolo_weights", None)
        if yw is None and getattr(model_config, "yolo_weights_path", None):
            with open(model_config.yolo_weights_path, "rb") as f:
                yw = f.read()
        if yw is None and getattr(model_config, "yolo_synthetic", False):
            from ... import yolov4
            yw = yolov4.make_state(getattr(model_config, "weights_seed", 0))
        if yw is not None and self.detector is None:
            from ...det_engine import DetEngine
            self.det = DetEngine(cam_config.width, cam_config.height, device=getattr(model_config, "device", 0), max_batch=1)
            self.det.load_weights(yw)
            self.detector = lambda frame: self.det.forward(np.asarray(frame, dtype=np.uint8)[None])

        self.engine = None
        if not self.just_box or self.detector is not None:
            self.engine = HpeEngine(cam_config.fx, cam_config.fy, cam_config.ppx, cam_config.ppy,
                                    cam_config.width, cam_config.height,
                                    device=getattr(model_config, "device", 0),
                                    max_batch=getattr(model_config, "max_batch", 64))
            w = getattr(model_config, "weights", None)
            if w is None and getattr(model_config, "weights_path", None):
                with open(model_config.weights_path, "rb") as f:
                    w = f.read()
            if w is None:   # no MetrABS export available (reference .gitignore:12-18): synthetic weights
                w = effnetv2.make_state(getattr(model_config, "weights_seed", 0))
            self.engine.load_weights(w)
            self.engine.set_joint_map(self.expand_joints, indices)
            if self.num_aug > 0:
                # hpe.py:88-100: n_test crops per frame (engine.crop_params / engine.warp). estimate() then raises like
                # the reference's does -- its decode takes one sample (hpe.py:108) -- with the library's explanation
                self.engine.set_augmentations(self.num_aug)

    def _bbox(self, frame):
        if self.detector is not None:                     # hpe.py:59-73, post-processing on the GPU
            boxes, confs = self.detector(frame)
            bbox, found = self.engine.select_person(boxes, confs, self.yolo_thresh)
            return tuple(int(v) for v in bbox[0]) if found[0] else None
        if self.bbox_provider is not None:
            return self.bbox_provider(frame)
        if self.fixed_bbox is not None:
            return self.fixed_bbox
        return 0, frame.shape[1] - 1, 0, frame.shape[0] - 1

    def estimate(self, frame):
        box = self._bbox(frame)
        if box is None:                                   # no human found (hpe.py:72-73)
            return None
        x1, x2, y1, y2 = (int(v) if int(v) > 0 else 0 for v in box)      # hpe.py:76-79
        if self.just_box:                                 # hpe.py:82-83 (note the different order)
            return {"bbox": (x1, y1, x2, y2)}
        joints, valid = self.engine.forward(np.asarray(frame, dtype=np.uint8)[None],
                                            np.array([[x1, x2, y1, y2]], np.int32))
        if not valid[0]:                                  # < 25 % of the joints in the FOV (hpe.py:152-153)
            return None
        return {"pose": joints[0].astype(np.float64),
                "edges": self.edges,
                "bbox": (x1, x2, y1, y2)}
