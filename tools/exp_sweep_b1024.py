"""do 256-row tiles win once M is large enough?  **[probes]**
The SE-gated projections and the expand GEMMs of the 16 x 16 and 8 x 8 stages at B = 256 and B = 1024 frames (M = 65 536 / 262 144 and
16 384 / 65 536 rows), every tile form the probe build has for them: the 128-row tile kernels the product selects (141 / 143 / 144 / 146),
the loader-wave kernels (155 / 156), the 256-row tiles (152 / 153: gated, 8 waves; 133 / 139: ungated 256 x 192 / 256 x 256).
    ISB_BUILD_PROBES=1 python -m isbfsar_amd.build --force;  PYTHONPATH=. python tools/exp_sweep_b1024.py"""
import sys

import numpy as np

from isbfsar_amd.hpe_engine import conv_debug, f32_to_bf16, f32_to_f16

F16 = "--bf16" not in sys.argv        # the 256-row probe tiles (152 / 153 / 133 / 139) exist in bf16 only: --bf16 runs every variant on bf16 operands

LAYERS = [  # name, HW, Cin, Cout, gate + residual, variants
    ("proj(SE) 1344->224 @16", 16, 1344, 224, 1, [0, 143, 141, 152, 153]),
    ("proj(SE) 768->192 @16", 16, 768, 192, 1, [0, 141, 152, 153]),
    ("proj(SE) 2304->384 @8", 8, 2304, 384, 1, [0, 141, 146, 155, 152, 153]),
    ("proj(SE) 3840->640 @8", 8, 3840, 640, 1, [0, 144, 146, 156, 152, 153]),
    ("exp1x1 640->3840 @8", 8, 640, 3840, 0, [0, 131, 132, 133, 139]),
]
rng = np.random.default_rng(0)
for B in (256, 1024):
    for name, hw, cin, cout, g, variants in LAYERS:
        cvt = f32_to_f16 if F16 else f32_to_bf16
        x = cvt(rng.normal(0, 1, (B, hw, hw, cin)).astype(np.float32))
        w = (rng.normal(0, 1, (cout, 1, 1, cin)) / np.sqrt(cin)).astype(np.float32)
        sc = np.ones(cout, np.float32)
        sh = np.zeros(cout, np.float32)
        res = cvt(rng.normal(0, 1, (B, hw, hw, cout)).astype(np.float32)) if g else None
        gate = rng.uniform(0.1, 0.9, (B, cin)).astype(np.float32) if g else None
        fl = 2.0 * B * hw * hw * cin * cout
        row = []
        for v in variants:
            try:
                _, ms = conv_debug(x, w, sc, sh, 1, 1, 0 if g else 1, res, gate, variant=v, iters=20, f16=F16)
                row.append(f"v{v}: {ms * 1e3:7.1f} us {fl / ms / 1e9:5.0f} TF")
            except Exception as e:
                row.append(f"v{v}: n/a ({str(e)[-40:]})")
        print(f"B={B:4d} {name:24s} " + " | ".join(row), flush=True)
