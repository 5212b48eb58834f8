"""Is a layer held back by the clock the chip sustains under its load? Same kernel, random against all-zero operands
(MI355X_MICROARCH.md, DVFS give-back (1)): equal times = not power-limited."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from isbfsar_amd.hpe_engine import conv_debug, f32_to_bf16
B = 256
rng = np.random.default_rng(0)
for name, hw, cin, cout, gated, act in [("proj 1344->224 @16", 16, 1344, 224, True, 0), ("proj 2304->384 @8", 8, 2304, 384, True, 0),
                                        ("expand 224->1344 @16", 16, 224, 1344, False, 1), ("expand 640->3840 @8", 8, 640, 3840, False, 1)]:
    for zero in (False, True, False, True):
        x = np.zeros((B, hw, hw, cin), np.float32) if zero else rng.normal(0, 1, (B, hw, hw, cin)).astype(np.float32)
        w = np.zeros((cout, 1, 1, cin), np.float32) if zero else (rng.normal(0, 1, (cout, 1, 1, cin)) / np.sqrt(cin)).astype(np.float32)
        sc = np.ones(cout, np.float32); sh = np.zeros(cout, np.float32)
        res = f32_to_bf16(np.zeros((B, hw, hw, cout), np.float32)) if gated else None
        gate = (np.zeros((B, cin), np.float32) if zero else rng.uniform(0.1, 0.9, (B, cin)).astype(np.float32)) if gated else None
        xb = f32_to_bf16(x)
        for _ in range(3):
            conv_debug(xb, w, sc, sh, 1, 1, act, res, gate, variant=0, iters=20)
        _, ms = conv_debug(xb, w, sc, sh, 1, 1, act, res, gate, variant=0, iters=50)
        print(f"{name:22s} {'zeros ' if zero else 'random'} {ms*1e3:7.1f} us", flush=True)
