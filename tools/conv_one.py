"""Run ONE conv layer/variant a few times (for rocprofv3 --pmc): python tools/conv_one.py hw cin cout k variant [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from isbfsar_amd.hpe_engine import conv_debug, f32_to_bf16
hw, cin, cout, k, v = [int(a) for a in sys.argv[1:6]]
B = int(sys.argv[6]) if len(sys.argv) > 6 else 256
rng = np.random.default_rng(0)
x = f32_to_bf16(rng.normal(0, 1, (B, hw, hw, cin)).astype(np.float32))
w = (rng.normal(0, 1, (cout, k, k, cin)) / np.sqrt(k * k * cin)).astype(np.float32)
_, ms = conv_debug(x, w, np.ones(cout, np.float32), np.zeros(cout, np.float32), k, 1, 1, None, None, variant=v, iters=3)
print(f"v{v} {ms*1e3:.1f} us")
