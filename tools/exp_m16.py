"""32x32x16 against 16x16x32 MFMA in the pipelined expand kernel under SUSTAINED load (the clock the chip holds depends on the
load of the last seconds): long alternating runs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from isbfsar_amd.hpe_engine import conv_debug, f32_to_bf16
B = 256
rng = np.random.default_rng(0)
for name, hw, cin, cout, va, vb in [("224->1344 @16", 16, 224, 1344, 184, 187), ("384->2304 @8", 8, 384, 2304, 186, 188), ("192->768 @16", 16, 192, 768, 184, 187)]:
    x = f32_to_bf16(rng.normal(0, 1, (B, hw, hw, cin)).astype(np.float32))
    w = (rng.normal(0, 1, (cout, 1, 1, cin)) / np.sqrt(cin)).astype(np.float32)
    sc = np.ones(cout, np.float32); sh = np.zeros(cout, np.float32)
    for rep in range(3):
        for v in (va, vb):
            _, ms = conv_debug(x, w, sc, sh, 1, 1, 1, None, None, variant=v, iters=2000)
            print(f"{name:16s} v{v} {ms*1e3:7.1f} us", flush=True)
