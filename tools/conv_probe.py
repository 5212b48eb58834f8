"""Tuning aid: where does a 1x1 expand conv spend its time? Vary K, the activation and the tile variant."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from isbfsar_amd.hpe_engine import conv_debug, f32_to_bf16

B = int(os.environ.get("SWEEP_B", "256"))
variants = [int(v) for v in os.environ.get("SWEEP_VARIANTS", "54,55,16,18,56").split(",")]
rng = np.random.default_rng(0)
CASES = [  # hw, cin, cout, k, act
    (8, 384, 2304, 1, 1), (8, 384, 2304, 1, 0), (8, 768, 2304, 1, 1), (8, 1536, 2304, 1, 1), (8, 3072, 2304, 1, 1),
    (8, 3072, 2304, 1, 0), (16, 224, 1344, 1, 1), (16, 224, 1344, 1, 0), (16, 896, 1344, 1, 1),
    (64, 64, 256, 3, 1), (64, 64, 256, 3, 0), (64, 256, 64, 1, 0),
]
for hw, cin, cout, k, act in CASES:
    x = f32_to_bf16(rng.normal(0, 1, (B, hw, hw, cin)).astype(np.float32))
    w = (rng.normal(0, 1, (cout, k, k, cin)) / np.sqrt(k * k * cin)).astype(np.float32)
    sc = np.ones(cout, np.float32); sh = np.zeros(cout, np.float32)
    fl = 2.0 * B * hw * hw * k * k * cin * cout
    row = []
    for v in variants:
        try:
            _, ms = conv_debug(x, w, sc, sh, k, 1, act, None, None, variant=v, iters=5)
            row.append(f"v{v}:{fl / ms / 1e9:5.0f}TF {ms*1e3:6.1f}us")
        except Exception as e:
            row.append(f"v{v}:  err")
    print(f"hw{hw} {cin}->{cout} k{k} act{act}  " + " ".join(row), flush=True)
