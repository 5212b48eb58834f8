"""Loader-wave 256 x 224 tile (variant 157) on the 16x16 stage's 1344 / 1152 -> 224 projections against the default pick."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from isbfsar_amd.hpe_engine import conv_debug, f32_to_bf16
for B, Cin in ((256, 1344), (128, 1344), (256, 1152), (64, 1344)):
    Cout = 224
    rng = np.random.default_rng(1)
    x = f32_to_bf16(rng.normal(0, 1, (B, 16, 16, Cin)).astype(np.float32))
    w = (rng.normal(0, 1, (Cout, 1, 1, Cin)) / np.sqrt(Cin)).astype(np.float32)
    sc = rng.uniform(0.8, 1.2, Cout).astype(np.float32); sh = rng.normal(0, 0.1, Cout).astype(np.float32)
    res = f32_to_bf16(rng.normal(0, 1, (B, 16, 16, Cout)).astype(np.float32))
    gate = rng.uniform(0.1, 0.9, (B, Cin)).astype(np.float32)
    fl = 2.0 * B * 256 * Cin * Cout
    out = []
    ref = None
    for v in (0, 143, 157):
        o, ms = conv_debug(x, w, sc, sh, 1, 1, 0, res, gate, variant=v, iters=200)
        if ref is None: ref = o
        out.append(f"v{v}: {ms * 1e3:6.1f} us {fl / ms / 1e9:5.0f} TF/s same={np.array_equal(o, ref)}")
    print(f"B={B} {Cin}->{Cout} @16: " + " | ".join(out), flush=True)
