"""Tuning aid: 64-wide k-tile variants against the current choices (variant 0) on the layers that carry the time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from isbfsar_amd.hpe_engine import conv_debug, f32_to_bf16
B = int(os.environ.get("SWEEP_B", "256"))
rng = np.random.default_rng(0)
U = [0, 101, 102, 103, 104, 106, 107, 108]
G = [0, 111, 112, 113, 114, 115, 116]
CASES = [  # hw, cin, cout, k, act, res, gate, variants
    (8, 384, 2304, 1, 1, 0, 0, U), (8, 640, 3840, 1, 1, 0, 0, U), (16, 192, 768, 1, 1, 0, 0, U), (8, 640, 1280, 1, 1, 0, 0, U),
    (64, 64, 256, 3, 1, 0, 0, U), (64, 256, 64, 1, 0, 1, 0, U), (64, 128, 64, 1, 0, 0, 0, U), (32, 384, 96, 1, 0, 1, 0, U),
    (8, 2304, 384, 1, 0, 1, 1, G), (8, 3840, 640, 1, 0, 1, 1, G), (16, 1344, 224, 1, 0, 1, 1, G), (16, 768, 192, 1, 0, 1, 1, G),
]
for hw, cin, cout, k, act, r, g, vs in CASES:
    x = f32_to_bf16(rng.normal(0, 1, (B, hw, hw, cin)).astype(np.float32))
    w = (rng.normal(0, 1, (cout, k, k, cin)) / np.sqrt(k * k * cin)).astype(np.float32)
    sc = np.ones(cout, np.float32); sh = np.zeros(cout, np.float32)
    res = f32_to_bf16(rng.normal(0, 1, (B, hw, hw, cout)).astype(np.float32)) if r else None
    gate = rng.uniform(0.1, 0.9, (B, cin)).astype(np.float32) if g else None
    fl = 2.0 * B * hw * hw * k * k * cin * cout
    row = []
    for v in vs:
        try:
            _, ms = conv_debug(x, w, sc, sh, k, 1, act, res, gate, variant=v, iters=5)
            row.append(f"v{v}:{fl / ms / 1e9:4.0f}TF/{ms*1e3:5.1f}us")
        except Exception as e:
            row.append(f"v{v}: err")
    print(f"hw{hw} {cin}->{cout} k{k} g{g}  " + " ".join(row), flush=True)
