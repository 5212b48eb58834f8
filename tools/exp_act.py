"""SiLU on/off and tile-variant timings of representative layers (B=256), HIP-event time per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from isbfsar_amd.hpe_engine import conv_debug, f32_to_bf16
B = 256
rng = np.random.default_rng(0)
def run(name, hw, cin, cout, k, variants, acts=(1,), res=False):
    x = f32_to_bf16(rng.normal(0, 1, (B, hw, hw, cin)).astype(np.float32))
    w = (rng.normal(0, 1, (cout, k, k, cin)) / np.sqrt(k * k * cin)).astype(np.float32)
    sc = np.ones(cout, np.float32); sh = np.zeros(cout, np.float32)
    r = f32_to_bf16(rng.normal(0, 1, (B, hw, hw, cout)).astype(np.float32)) if res else None
    fl = 2.0 * B * hw * hw * k * k * cin * cout
    for v in variants:
        for a in acts:
            try:
                conv_debug(x, w, sc, sh, k, 1, a, r, None, variant=v, iters=5)
                _, ms = conv_debug(x, w, sc, sh, k, 1, a, r, None, variant=v, iters=20)
                print(f"{name:28s} v{v:<4d} act={a} {ms*1e3:7.1f} us {fl/ms/1e9:6.0f} TF/s", flush=True)
            except Exception as e:
                print(f"{name:28s} v{v} act={a} err {str(e)[:80]}", flush=True)
run("exp1x1 192->768 @16", 16, 192, 768, 1, [0], (1, 0))
run("exp1x1 224->1344 @16", 16, 224, 1344, 1, [0, 133], (1, 0))
run("exp1x1 384->2304 @8", 8, 384, 2304, 1, [0, 133], (1, 0))
run("exp1x1 640->3840 @8", 8, 640, 3840, 1, [0, 133], (1, 0))
run("proj 384->96 @32 (res)", 32, 384, 96, 1, [0, 132, 136], (0,), res=True)
run("head 640->1280 @8", 8, 640, 1280, 1, [0, 131, 133], (1,))
