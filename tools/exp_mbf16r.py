"""Round 6: the fused fronts of the 8 x 8 and 16 x 16 MBConv blocks, one layer alone -- two launches (form 0), the round-5 kernel (1) and the
producer / consumer kernel (2), interleaved in one session, sustained launches.
    PYTHONPATH=. python tools/exp_mbf16r.py [B=256] [iters=200] [rounds=3]"""
import sys

import numpy as np

from isbfsar_amd.hpe_engine import f32_to_f16, mbfront_debug

args = [a for a in sys.argv[1:] if not a.startswith("--")]
B = int(args[0]) if len(args) > 0 else 256
iters = int(args[1]) if len(args) > 1 else 200
rounds = int(args[2]) if len(args) > 2 else 3
for hw, cin, cexp in ((8, 384, 2304), (16, 224, 1344), (16, 192, 1152), (16, 192, 768)):
    rng = np.random.default_rng(cin + cexp)
    x16 = f32_to_f16(rng.normal(0, 1, (B, hw, hw, cin)).astype(np.float32))
    w1 = (rng.normal(0, 1, (cexp, cin)) / np.sqrt(cin)).astype(np.float32)
    s1 = rng.uniform(0.8, 1.2, cexp).astype(np.float32)
    b1 = rng.uniform(-0.2, 0.2, cexp).astype(np.float32)
    dww = (rng.normal(0, 1, (cexp, 3, 3)) / 3.0).astype(np.float32)
    s2 = rng.uniform(0.8, 1.2, cexp).astype(np.float32)
    b2 = rng.uniform(-0.1, 0.1, cexp).astype(np.float32)
    ref = None
    for rd in range(rounds):
        row = []
        for form in (0, 1, 2):
            d, pl, ms = mbfront_debug(x16, w1, s1, b1, dww, s2, b2, f16=True, form=form, iters=iters)
            if ref is None:
                ref = (d, pl)
            same = bool(np.array_equal(d, ref[0]) and np.array_equal(pl, ref[1]))
            row.append(f"form {form}: {ms * 1e3:7.1f} us{'' if same else ' (BITS DIFFER)'}")
        print(f"{hw:2d}x{hw:<2d} {cin:3d} -> {cexp:4d}, {B} frames, {iters} launches: " + " | ".join(row), flush=True)
    if (cin == 224 or hw == 8) and "--stamps" in sys.argv:
        mbfront_debug(x16, w1, s1, b1, dww, s2, b2, f16=True, form=2 | 0x100, iters=20)
