#!/usr/bin/env python3
"""Single-frame latency of the drop-in HumanPoseEstimator.estimate() (reference modules/hpe/hpe.py:48-173, the call
main.py:336-342 makes per camera frame): a pageable numpy frame in, a pose dict out, synchronous. With a fixed box (the
pose stage alone) and with the built-in YOLOv4 detector in front (detector -> box selection -> pose).

    python tools/estimate_latency.py [--iters 200] > profiles/rNN_estimate_latency.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from isbfsar_amd import synth                                        # noqa: E402
from isbfsar_amd.modules.hpe.hpe import HumanPoseEstimator          # noqa: E402
from isbfsar_amd.params import MetrabsTRTConfig, RealSenseIntrinsics  # noqa: E402


def measure(est, frame, iters):
    for _ in range(10):
        out = est.estimate(frame)
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        out = est.estimate(frame)
        ts.append(time.perf_counter() - t0)
    ts = np.sort(np.array(ts)) * 1e3
    return {"p50_ms": round(float(ts[len(ts) // 2]), 4), "p99_ms": round(float(ts[int(len(ts) * 0.99)]), 4),
            "mean_ms": round(float(ts.mean()), 4), "returned": None if out is None else sorted(out)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    a = ap.parse_args()
    frame = synth.frames(1, seed=0)[0]
    cam = RealSenseIntrinsics()
    res = {}
    cfg = MetrabsTRTConfig()
    cfg.fixed_bbox = tuple(int(v) for v in synth.bboxes(1, seed=0)[0])
    cfg.max_batch = 1
    res["pose_only_fixed_box"] = measure(HumanPoseEstimator(cfg, cam), frame, a.iters)
    cfg2 = MetrabsTRTConfig()
    cfg2.yolo_synthetic = True
    cfg2.max_batch = 1
    est2 = HumanPoseEstimator(cfg2, cam)
    r = measure(est2, frame, a.iters)
    r["note"] = "synthetic detector weights on a noise frame: when no person box clears the threshold estimate() returns None after the detector alone"
    res["detector_plus_pose"] = r
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
