"""Loader-wave projection kernel on the 16x16 stage's 768 -> 192 projection (tile kernel 141 is the default there)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from isbfsar_amd.hpe_engine import conv_debug, f32_to_bf16
for B in (256, 128):
    Cin, Cout = 768, 192
    rng = np.random.default_rng(1)
    x = f32_to_bf16(rng.normal(0, 1, (B, 16, 16, Cin)).astype(np.float32))
    w = (rng.normal(0, 1, (Cout, 1, 1, Cin)) / np.sqrt(Cin)).astype(np.float32)
    sc = np.ones(Cout, np.float32); sh = np.zeros(Cout, np.float32)
    res = f32_to_bf16(rng.normal(0, 1, (B, 16, 16, Cout)).astype(np.float32))
    gate = rng.uniform(0.1, 0.9, (B, Cin)).astype(np.float32)
    fl = 2.0 * B * 256 * Cin * Cout
    out = []
    ref = None
    for v in (141, 146, 155):
        o, ms = conv_debug(x, w, sc, sh, 1, 1, 0, res, gate, variant=v, iters=200)
        if ref is None: ref = o
        out.append(f"v{v}: {ms * 1e3:6.1f} us {fl / ms / 1e9:5.0f} TF/s same={np.array_equal(o, ref)}")
    print(f"B={B} 768->192 @16: " + " | ".join(out), flush=True)
