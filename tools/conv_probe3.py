"""Tuning aid: gated projection shapes, with and without the SE gate, register-staged vs LDS-DMA tiles."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from isbfsar_amd.hpe_engine import conv_debug, f32_to_bf16
B = int(os.environ.get("SWEEP_B", "256"))
rng = np.random.default_rng(0)
for hw, cin, cout in [(8, 2304, 384), (8, 3840, 640), (16, 1344, 224), (16, 768, 192)]:
    x = f32_to_bf16(rng.normal(0, 1, (B, hw, hw, cin)).astype(np.float32))
    w = (rng.normal(0, 1, (cout, 1, 1, cin)) / np.sqrt(cin)).astype(np.float32)
    sc = np.ones(cout, np.float32); sh = np.zeros(cout, np.float32)
    res = f32_to_bf16(rng.normal(0, 1, (B, hw, hw, cout)).astype(np.float32))
    gate = rng.uniform(0.1, 0.9, (B, cin)).astype(np.float32)
    fl = 2.0 * B * hw * hw * cin * cout
    for g, vs in [(gate, [0, 141, 142, 143, 144, 145, 146, 147, 148]), (None, [131, 132])]:
        row = []
        for v in vs:
            try:
                _, ms = conv_debug(x, w, sc, sh, 1, 1, 0, res, g, variant=v, iters=5)
                row.append(f"v{v}:{fl / ms / 1e9:4.0f}TF/{ms*1e3:5.1f}us")
            except Exception as e:
                row.append(f"v{v}: err")
        print(f"hw{hw} {cin}->{cout} gate={'y' if g is not None else 'n'}  " + " ".join(row), flush=True)
