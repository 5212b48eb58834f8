"""How a lean 1x1 GEMM launch splits into a fixed part (prologue + epilogue) and a per-k-step part: the same M x N
at growing K, least-squares line through (k-steps, time)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from isbfsar_amd.hpe_engine import conv_debug, f32_to_bf16

B = int(os.environ.get("PROBE_B", "128"))
HW, COUT = int(os.environ.get("PROBE_HW", "16")), int(os.environ.get("PROBE_COUT", "1344"))
VAR = int(os.environ.get("PROBE_VARIANT", "131"))
ACT = int(os.environ.get("PROBE_ACT", "1"))
rng = np.random.default_rng(0)
pts = []
for cin in [int(v) for v in os.environ.get("PROBE_CINS", "224,448,896,1792").split(",")]:
    x = f32_to_bf16(rng.normal(0, 1, (B, HW, HW, cin)).astype(np.float32))
    w = (rng.normal(0, 1, (COUT, 1, 1, cin)) / np.sqrt(cin)).astype(np.float32)
    sc = np.ones(COUT, np.float32); sh = np.zeros(COUT, np.float32)
    _, ms = conv_debug(x, w, sc, sh, 1, 1, ACT, None, None, variant=VAR, iters=10)
    _, ms = conv_debug(x, w, sc, sh, 1, 1, ACT, None, None, variant=VAR, iters=10)
    pts.append((cin // 32, ms))
    print(f"Cin={cin:5d} k-steps={cin // 32:3d}  {ms * 1e3:8.1f} us  {2.0 * B * HW * HW * cin * COUT / ms / 1e9:6.0f} TFLOP/s", flush=True)
k = np.array([p[0] for p in pts], float); t = np.array([p[1] for p in pts]) * 1e3
a, b = np.polyfit(k, t, 1)
M = B * HW * HW
wgs = -(-M // 128) * -(-COUT // 192)
print(f"fit: {b:.1f} us fixed + {a:.2f} us per k-step; {wgs} workgroups of 128x192 -> per k-step and workgroup round "
      f"({wgs / 512:.1f} rounds of 512 slots): {a / (wgs / 512) * 1e3:.0f} ns; MFMA-bound floor 160 ns")
