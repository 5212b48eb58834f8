"""Tuning aid: time representative EfficientNetV2-L conv shapes for every tile variant (isb_debug_conv)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from isbfsar_amd.hpe_engine import conv_debug, f32_to_bf16

B = int(os.environ.get("SWEEP_B", "128"))
LAYERS = [  # name, HW, Cin, Cout, k, stride, gate, res
    ("f3x3 32->32 @128", 128, 32, 32, 3, 1, 0, 1),
    ("exp3x3 64->256 @64", 64, 64, 256, 3, 1, 0, 0),
    ("proj 256->64 @64", 64, 256, 64, 1, 1, 0, 1),
    ("exp3x3 96->384 @32", 32, 96, 384, 3, 1, 0, 0),
    ("proj 384->96 @32", 32, 384, 96, 1, 1, 0, 1),
    ("exp1x1 192->768 @16", 16, 192, 768, 1, 1, 0, 0),
    ("proj 768->192 @16", 16, 768, 192, 1, 1, 1, 1),
    ("exp1x1 224->1344 @16", 16, 224, 1344, 1, 1, 0, 0),
    ("proj 1344->224 @16", 16, 1344, 224, 1, 1, 1, 1),
    ("exp1x1 384->2304 @8", 8, 384, 2304, 1, 1, 0, 0),
    ("proj 2304->384 @8", 8, 2304, 384, 1, 1, 1, 1),
    ("exp1x1 640->3840 @8", 8, 640, 3840, 1, 1, 0, 0),
    ("proj 3840->640 @8", 8, 3840, 640, 1, 1, 1, 1),
]
variants = [int(v) for v in os.environ.get("SWEEP_VARIANTS", "0,1,2,3,4,5,6,7,8,9").split(",")]
only = os.environ.get("SWEEP_ONLY")
rng = np.random.default_rng(0)
print(f"B={B}")
for name, hw, cin, cout, k, s, g, r in LAYERS:
    if only and only not in name:
        continue
    x = f32_to_bf16(rng.normal(0, 1, (B, hw, hw, cin)).astype(np.float32))
    w = (rng.normal(0, 1, (cout, k, k, cin)) / np.sqrt(k * k * cin)).astype(np.float32)
    sc = np.ones(cout, np.float32); sh = np.zeros(cout, np.float32)
    oh = hw // s
    res = f32_to_bf16(rng.normal(0, 1, (B, oh, oh, cout)).astype(np.float32)) if r else None
    gate = rng.uniform(0.1, 0.9, (B, cin)).astype(np.float32) if g else None
    fl = 2.0 * B * oh * oh * k * k * cin * cout
    row = []
    for v in variants:
        try:
            _, ms = conv_debug(x, w, sc, sh, k, s, 1 - r, res, gate, variant=v, iters=5)
            row.append(f"v{v}:{fl / ms / 1e9:6.0f}")
        except Exception as e:
            row.append(f"v{v}:  err")
    print(f"{name:24s} " + " ".join(row), flush=True)
