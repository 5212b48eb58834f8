"""3x3 tile variants on the detector's stride-1 3x3 layers at 256 frames (LeakyReLU / Mish epilogues as in the network)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from isbfsar_amd.hpe_engine import conv_debug, f32_to_bf16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for hw, cin, cout, act in ((128, 32, 64, 2), (64, 64, 64, 2), (32, 128, 128, 2), (32, 128, 256, 3), (16, 256, 256, 2), (16, 256, 512, 3), (8, 512, 512, 2), (8, 512, 1024, 3)):
    rng = np.random.default_rng(hw + cin)
    x = f32_to_bf16(rng.normal(0, 1, (B, hw, hw, cin)).astype(np.float32))
    w = (rng.normal(0, 1, (cout, 3, 3, cin)) / np.sqrt(9 * cin)).astype(np.float32)
    sc = np.ones(cout, np.float32); sh = np.zeros(cout, np.float32)
    fl = 2.0 * B * hw * hw * 9 * cin * cout
    out, ref = [], None
    for v in (162, 161, 164, 165, 166, 168):
        if v in (161, 166) and cout % 192: continue
        if v == 165 and cout != 64: continue
        try:
            o, ms = conv_debug(x, w, sc, sh, 3, 1, act, None, None, variant=v, iters=20)
            if ref is None: ref = o
            out.append(f"v{v}: {ms * 1e3:7.1f} us {fl / ms / 1e9:5.0f} TF/s{'' if np.array_equal(o, ref) else ' DIFF'}")
        except Exception as e:
            out.append(f"v{v}: {str(e)[:30]}")
    print(f"{hw:3d} {cin}->{cout}: " + " | ".join(out), flush=True)
