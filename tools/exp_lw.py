"""Loader-wave projection kernel (variants 155 / 156) against the tile kernels on the 8x8 stages' shapes, 256 and 128 frames."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from isbfsar_amd.hpe_engine import conv_debug, f32_to_f16
for B in (256, 128):
    for Cin, Cout, cands in ((2304, 384, (146, 141, 155)), (3840, 640, (144, 146, 156, 155)), (1344, 384, (146, 155))):
        rng = np.random.default_rng(1)
        x = f32_to_f16(rng.normal(0, 1, (B, 8, 8, Cin)).astype(np.float32))
        w = (rng.normal(0, 1, (Cout, 1, 1, Cin)) / np.sqrt(Cin)).astype(np.float32)
        sc = np.ones(Cout, np.float32); sh = np.zeros(Cout, np.float32)
        res = f32_to_f16(rng.normal(0, 1, (B, 8, 8, Cout)).astype(np.float32))
        gate = rng.uniform(0.1, 0.9, (B, Cin)).astype(np.float32)
        fl = 2.0 * B * 64 * Cin * Cout
        out = []
        for v in cands:
            try:
                _, ms = conv_debug(x, w, sc, sh, 1, 1, 0, res, gate, variant=v, iters=200, f16=True)
                out.append(f"v{v}: {ms * 1e3:6.1f} us {fl / ms / 1e9:5.0f} TF/s")
            except Exception as e:
                out.append(f"v{v}: {str(e)[:40]}")
        print(f"B={B} {Cin}->{Cout}: " + " | ".join(out), flush=True)
