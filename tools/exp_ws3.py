import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from isbfsar_amd.hpe_engine import conv_debug, f32_to_bf16
B = 256
rng = np.random.default_rng(0)
hw, cin, cout = [int(v) for v in os.environ.get("EXP_LAYER", "16,224,1344").split(",")]
x = f32_to_bf16(rng.normal(0, 1, (B, hw, hw, cin)).astype(np.float32))
w = (rng.normal(0, 1, (cout, 1, 1, cin)) / np.sqrt(cin)).astype(np.float32)
sc = np.ones(cout, np.float32); sh = np.zeros(cout, np.float32)
mode = sys.argv[1] if len(sys.argv) > 1 else "stamp"
if mode == "stamp":
    _, ms = conv_debug(x, w, sc, sh, 1, 1, 1, None, None, variant=900000 + int(os.environ.get('EXP_V', '181')), iters=5)
    print(f"stamped v181 {ms*1e3:7.1f} us", flush=True)
else:
    for act in (1, 0):
        conv_debug(x, w, sc, sh, 1, 1, act, None, None, variant=181, iters=5)
        _, ms = conv_debug(x, w, sc, sh, 1, 1, act, None, None, variant=181, iters=20)
        print(f"v181 act={act} {ms*1e3:7.1f} us", flush=True)
