"""In-kernel phase clocks of one tile-GEMM layer (B=256): python tools/exp_stamp.py hw cin cout variant [gated] [res]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from isbfsar_amd.hpe_engine import conv_debug, f32_to_bf16
hw, cin, cout, v = [int(a) for a in sys.argv[1:5]]
gated = len(sys.argv) > 5 and sys.argv[5] == "1"
B = 256
rng = np.random.default_rng(0)
x = f32_to_bf16(rng.normal(0, 1, (B, hw, hw, cin)).astype(np.float32))
w = (rng.normal(0, 1, (cout, 1, 1, cin)) / np.sqrt(cin)).astype(np.float32)
sc = np.ones(cout, np.float32); sh = np.zeros(cout, np.float32)
res = f32_to_bf16(rng.normal(0, 1, (B, hw, hw, cout)).astype(np.float32)) if gated else None
gate = rng.uniform(0.1, 0.9, (B, cin)).astype(np.float32) if gated else None
_, ms = conv_debug(x, w, sc, sh, 1, 1, 0 if gated else 1, res, gate, variant=900000 + v, iters=5)
print(f"{hw} {cin}->{cout} v{v} stamped: {ms*1e3:.1f} us", flush=True)
