"""Gated projections (stage 4-7 shapes, B=256): lean 2-buffer kernels (14x) vs the 4-buffer ring (8x) vs 64-wide k (11x)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from isbfsar_amd.hpe_engine import conv_debug, f32_to_bf16
B = 256
rng = np.random.default_rng(0)
for name, hw, cin, cout, variants in [("proj 768->192 @16", 16, 768, 192, [0, 141, 191]),
                                      ("proj 1344->224 @16", 16, 1344, 224, [0, 143, 193]),
                                      ("proj 2304->384 @8", 8, 2304, 384, [0, 146, 196]),
                                      ("proj 3840->640 @8", 8, 3840, 640, [0, 144, 194])]:
    x = f32_to_bf16(rng.normal(0, 1, (B, hw, hw, cin)).astype(np.float32))
    w = (rng.normal(0, 1, (cout, 1, 1, cin)) / np.sqrt(cin)).astype(np.float32)
    sc = np.ones(cout, np.float32); sh = np.zeros(cout, np.float32)
    res = f32_to_bf16(rng.normal(0, 1, (B, hw, hw, cout)).astype(np.float32))
    gate = rng.uniform(0.1, 0.9, (B, cin)).astype(np.float32)
    fl = 2.0 * B * hw * hw * cin * cout
    for v in variants:
        for g in ((gate, "gated"), (None, "plain")) if v == 0 else ((gate, "gated"),):
            try:
                conv_debug(x, w, sc, sh, 1, 1, 0, res, g[0], variant=v, iters=3)
                _, ms = conv_debug(x, w, sc, sh, 1, 1, 0, res, g[0], variant=v, iters=10)
                print(f"{name:22s} v{v:<4d} {g[1]} {ms*1e3:7.1f} us {fl/ms/1e9:6.0f} TF/s", flush=True)
            except Exception as e:
                print(f"{name:22s} v{v} err {str(e)[:90]}", flush=True)
