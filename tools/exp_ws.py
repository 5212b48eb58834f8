"""variant 181 (weights-stationary) vs the tile kernel on the expand layers, B=256, HIP-event time per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from isbfsar_amd.hpe_engine import conv_debug, f32_to_bf16
B = int(os.environ.get("EXP_B", "256"))
rng = np.random.default_rng(0)
for name, hw, cin, cout in [("96->384 @32", 32, 96, 384), ("192->768 @16", 16, 192, 768), ("192->1152 @16", 16, 192, 1152),
                            ("224->1344 @16", 16, 224, 1344), ("384->2304 @8", 8, 384, 2304)]:
    x = f32_to_bf16(rng.normal(0, 1, (B, hw, hw, cin)).astype(np.float32))
    w = (rng.normal(0, 1, (cout, 1, 1, cin)) / np.sqrt(cin)).astype(np.float32)
    sc = np.ones(cout, np.float32); sh = np.zeros(cout, np.float32)
    fl = 2.0 * B * hw * hw * cin * cout
    for v in ((131, 184, 131, 184) if cin != 384 else (131, 185, 186, 131, 185)):
        conv_debug(x, w, sc, sh, 1, 1, 1, None, None, variant=v, iters=5)
        _, ms = conv_debug(x, w, sc, sh, 1, 1, 1, None, None, variant=v, iters=20)
        print(f"{name:16s} v{v} {ms*1e3:7.1f} us {fl/ms/1e9:6.0f} TF/s", flush=True)
