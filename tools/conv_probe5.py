"""Tuning aid: persistent pipelined variants against the current choices on un-gated, residual-free layers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from isbfsar_amd.hpe_engine import conv_debug, f32_to_bf16
B = int(os.environ.get("SWEEP_B", "256"))
rng = np.random.default_rng(0)
V = [0, 161, 162, 163, 164, 165, 166]
CASES = [(64, 64, 256, 3), (32, 96, 384, 3), (128, 32, 32, 3), (128, 32, 128, 3), (64, 64, 256, 30), (32, 96, 384, 30)]
for hw, cin, cout, k in CASES:
    st = 2 if (hw == 128 and cout == 128) or k == 30 else 1
    k = 3 if k == 30 else k
    x = f32_to_bf16(rng.normal(0, 1, (B, hw, hw, cin)).astype(np.float32))
    w = (rng.normal(0, 1, (cout, k, k, cin)) / np.sqrt(k * k * cin)).astype(np.float32)
    sc = np.ones(cout, np.float32); sh = np.zeros(cout, np.float32)
    fl = 2.0 * B * (hw // st) ** 2 * k * k * cin * cout
    row = []
    for v in V:
        try:
            _, ms = conv_debug(x, w, sc, sh, k, st, 1, None, None, variant=v, iters=5)
            row.append(f"v{v}:{fl / ms / 1e9:4.0f}TF/{ms*1e3:5.1f}us")
        except Exception as e:
            row.append(f"v{v}: err")
    print(f"hw{hw} {cin}->{cout} k{k}  " + " ".join(row), flush=True)
