"""Seeded synthetic inputs for tests and bench.

* frames: ``uint8[480,640,3]`` BGR like ``RealSense.read()`` (reference ``utils/input.py``),
* bboxes: ``(x1, x2, y1, y2)`` in the order ``HumanPoseEstimator.estimate`` returns them
  (reference ``modules/hpe/hpe.py:173``) -- the detector is bypassed because noise frames
  contain no person,
* skeleton windows: root-centred random walks in the value range observed in the reference's
  ``assets/saved/support_set.pkl`` (the distribution ``main.py:102-105`` feeds to the AR module).
"""
from __future__ import annotations

import numpy as np


def frames(n: int, seed: int = 0, h: int = 480, w: int = 640) -> np.ndarray:
    out = np.empty((n, h, w, 3), np.uint8)
    for i in range(n):
        out[i] = np.random.default_rng(seed + i).integers(0, 256, (h, w, 3), dtype=np.uint8)
    return out


def bboxes(n: int, seed: int = 0, h: int = 480, w: int = 640, lo: int = 120, hi: int = 400) -> np.ndarray:
    """Uniform random boxes with sides in [lo, hi] fully inside the frame; int32[n,4] = x1,x2,y1,y2."""
    rng = np.random.default_rng(1_000_003 + seed)
    bw = rng.integers(lo, hi + 1, n)
    bh = rng.integers(lo, min(hi, h - 1) + 1, n)
    x1 = (rng.random(n) * (w - bw)).astype(np.int64)
    y1 = (rng.random(n) * (h - bh)).astype(np.int64)
    return np.stack([x1, x1 + bw, y1, y1 + bh], axis=1).astype(np.int32)


def skeleton_windows(n: int, seq_len: int, n_joints: int, seed: int = 0, step: float = 0.02) -> np.ndarray:
    """float32[n, L, 3J]: x_0 ~ U(-0.45, 0.55), x_t = x_{t-1} + N(0, step^2), joint 0 == 0."""
    rng = np.random.default_rng(7_000_001 + seed)
    x0 = rng.uniform(-0.45, 0.55, (n, 1, n_joints, 3))
    steps = rng.normal(0.0, step, (n, seq_len, n_joints, 3))
    steps[:, 0] = 0.0
    x = x0 + np.cumsum(steps, axis=1)
    x = x - x[:, :, :1, :]                       # root-centre every frame (main.py:103)
    return x.reshape(n, seq_len, n_joints * 3).astype(np.float32)


def yolo_outputs(n: int = 6, seed: int = 99):
    """Seeded detector outputs in the reference's YOLOv4 export layout (modules/hpe/hpe.py:60):
    boxes float32[n,4032,1,4] (x1,y1,x2,y2 normalised), confidences float32[n,4032,80].
    Frame 3 has no detection above the 0.3 threshold, frame 4 only non-person classes, frame 5
    negative x1 coordinates (clamped to 0, hpe.py:76)."""
    rng = np.random.default_rng(seed)
    c = rng.uniform(0.0, 1.0, (n, 4032, 2, 2)).astype(np.float32)
    lo, hi = c.min(axis=2), c.max(axis=2)
    boxes = np.concatenate([lo, hi], axis=-1)[:, :, None, :].astype(np.float32)
    confs = (rng.uniform(0.0, 0.25, (n, 4032, 80))).astype(np.float32)
    for i in range(n):
        for _ in range(12):
            a, cls = int(rng.integers(0, 4032)), int(rng.integers(0, 3))
            confs[i, a, cls] = np.float32(rng.uniform(0.31, 0.99))
    if n > 3:
        confs[3] = np.minimum(confs[3], np.float32(0.25))
    if n > 4:
        confs[4, :, 0] = np.minimum(confs[4, :, 0], np.float32(0.2))
    if n > 5:
        boxes[5, :, 0, 0] = -0.01
    return boxes, confs
