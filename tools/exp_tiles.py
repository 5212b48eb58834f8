"""Tile variants on the un-gated expand / projection GEMMs (B=256): 131 128x192 (8 waves), 140 64x192 (4 waves), 133 256x192."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from isbfsar_amd.hpe_engine import conv_debug, f32_to_bf16
B = 256
rng = np.random.default_rng(0)
for name, hw, cin, cout, variants, act, res in [("exp 192->768 @16", 16, 192, 768, [131, 140, 150], 1, 0), ("exp 224->1344 @16", 16, 224, 1344, [131, 140], 1, 0),
                                     ("exp 384->2304 @8", 8, 384, 2304, [131, 140, 150], 1, 0), ("exp 640->3840 @8", 8, 640, 3840, [131, 140], 1, 0),
                                     ("proj 384->96 @32 res", 32, 384, 96, [132, 150, 138], 0, 1), ("head 640->1280 @8", 8, 640, 1280, [132, 140, 150], 1, 0)]:
    x = f32_to_bf16(rng.normal(0, 1, (B, hw, hw, cin)).astype(np.float32))
    w = (rng.normal(0, 1, (cout, 1, 1, cin)) / np.sqrt(cin)).astype(np.float32)
    sc = np.ones(cout, np.float32); sh = np.zeros(cout, np.float32)
    r = f32_to_bf16(rng.normal(0, 1, (B, hw, hw, cout)).astype(np.float32)) if res else None
    fl = 2.0 * B * hw * hw * cin * cout
    for v in variants:
        try:
            conv_debug(x, w, sc, sh, 1, 1, act, r, None, variant=v, iters=5)
            _, ms = conv_debug(x, w, sc, sh, 1, 1, act, r, None, variant=v, iters=20)
            print(f"{name:22s} v{v:<4d} {ms*1e3:7.1f} us {fl/ms/1e9:6.0f} TF/s", flush=True)
        except Exception as e:
            print(f"{name:22s} v{v} err {str(e)[:80]}", flush=True)
