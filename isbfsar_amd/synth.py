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
