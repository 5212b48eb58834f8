"""Tuning aid: one-launch Fused-MBConv block vs the two-launch path at the network's shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from isbfsar_amd.hpe_engine import conv_debug, fused_mb_debug, f32_to_bf16
B = int(os.environ.get("SWEEP_B", "256"))
rng = np.random.default_rng(0)
for H, cin, cexp, cout2, stride, res in [(64, 64, 256, 64, 1, 1), (128, 32, 128, 64, 2, 0), (32, 96, 384, 96, 1, 1), (64, 64, 256, 96, 2, 0)]:
    x = f32_to_bf16(rng.normal(0, 1, (B, H, H, cin)).astype(np.float32))
    w1 = (rng.normal(0, 1, (cexp, 3, 3, cin)) / np.sqrt(9 * cin)).astype(np.float32)
    w2 = (rng.normal(0, 1, (cout2, cexp)) / np.sqrt(cexp)).astype(np.float32)
    o1, z1 = np.ones(cexp, np.float32), np.zeros(cexp, np.float32)
    o2, z2 = np.ones(cout2, np.float32), np.zeros(cout2, np.float32)
    OH = H // stride
    r = f32_to_bf16(rng.normal(0, 1, (B, OH, OH, cout2)).astype(np.float32)) if res else None
    _, tf = fused_mb_debug(x, w1, o1, z1, w2, o2, z2, r, stride, iters=5)
    e, t1 = conv_debug(x, w1, o1, z1, 3, stride, 1, None, None, variant=0, iters=5)
    _, t2 = conv_debug(e, w2.reshape(cout2, 1, 1, cexp), o2, z2, 1, 1, 0, r, None, variant=0, iters=5)
    print(f"H{H} {cin}->{cexp}->{cout2} s{stride}: fused {tf*1e3:7.1f} us   two-launch {t1*1e3:7.1f} + {t2*1e3:6.1f} = {(t1+t2)*1e3:7.1f} us", flush=True)
