"""1x1 tile variants on the detector's 1x1 layers at 256 frames."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from isbfsar_amd.hpe_engine import conv_debug, f32_to_bf16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
shapes = ((128, 64, 64), (128, 128, 64), (64, 64, 64), (64, 128, 64), (64, 128, 128), (32, 128, 128), (32, 256, 128), (32, 256, 256),
          (16, 256, 256), (16, 512, 256), (16, 512, 512), (8, 512, 512), (8, 1024, 512), (8, 1024, 1024), (8, 2048, 512))
for hw, cin, cout in shapes:
    rng = np.random.default_rng(hw + cin)
    x = f32_to_bf16(rng.normal(0, 1, (B, hw, hw, cin)).astype(np.float32))
    w = (rng.normal(0, 1, (cout, 1, 1, cin)) / np.sqrt(cin)).astype(np.float32)
    sc = np.ones(cout, np.float32); sh = np.zeros(cout, np.float32)
    fl = 2.0 * B * hw * hw * cin * cout
    out, ref = [], None
    for v in (0, 132, 134, 135, 139, 150, 138):
        if v == 135 and cout != 64: continue
        if v == 139 and cout % 256: continue
        try:
            o, ms = conv_debug(x, w, sc, sh, 1, 1, 2, None, None, variant=v, iters=20)
            if ref is None: ref = o
            out.append(f"v{v}: {ms * 1e3:6.1f} us {fl / ms / 1e9:4.0f}{'' if np.array_equal(o, ref) else ' DIFF'}")
        except Exception as e:
            out.append(f"v{v}: {str(e)[:24]}")
    print(f"{hw:3d} {cin}->{cout}: " + " | ".join(out), flush=True)
