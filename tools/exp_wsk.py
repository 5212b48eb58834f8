"""A/B of the stationary-weights gated projection (conv_wsk.hip, variant 157) against the tile kernels on the layers it was built
for, 256 frames: python tools/exp_wsk.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from isbfsar_amd.hpe_engine import conv_debug, f32_to_f16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rng = np.random.default_rng(0)
for cin, cout, base in ((768, 192, 141), (1152, 224, 143), (1344, 224, 143)):
    x = f32_to_f16(rng.normal(0, 1, (B, 16, 16, cin)).astype(np.float32))
    w = (rng.normal(0, 1, (cout, 1, 1, cin)) / np.sqrt(cin)).astype(np.float32)
    sc = np.ones(cout, np.float32); sh = np.zeros(cout, np.float32)
    res = f32_to_f16(rng.normal(0, 1, (B, 16, 16, cout)).astype(np.float32))
    gate = rng.uniform(0.1, 0.9, (B, cin)).astype(np.float32)
    for rep in range(2):
        for v in (base, 157):
            _, ms = conv_debug(x, w, sc, sh, 1, 1, 0, res, gate, variant=v, f16=True, iters=20)
            print(f"{cin}->{cout} variant {v}: {ms * 1e3:.1f} us", flush=True)
