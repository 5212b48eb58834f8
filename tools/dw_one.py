"""Run the depthwise kernel a few times (for rocprofv3 --pmc): python tools/dw_one.py H C [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from isbfsar_amd.hpe_engine import dwconv_debug, f32_to_bf16
H, C = int(sys.argv[1]), int(sys.argv[2])
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
rng = np.random.default_rng(0)
x = f32_to_bf16(rng.normal(0, 1, (B, H, H, C)).astype(np.float32))
w = (rng.normal(0, 1, (C, 3, 3)) / 3).astype(np.float32)
_, _, ms = dwconv_debug(x, w, np.ones(C, np.float32), np.zeros(C, np.float32), 1, iters=3)
print(f"dw {H}x{H}x{C} B={B}: {ms*1e3:.1f} us  {2*B*H*H*C*2/ms/1e9:.0f} GB/s algorithmic")
