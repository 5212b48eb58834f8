"""Phase clocks of the Fused-MBConv kernel (conv_fused_mb.hip, ConvArgs.probe & 2): stage-2 body block 64 -> 256 -> 64 on 64 x 64 maps,
B frames (default 128): python tools/exp_fmb_stamps.py [B]   -> the kernel prints the mean phase cycles of workgroups 256 .. 319 to stderr"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from isbfsar_amd.hpe_engine import f32_to_f16, fused_mb_debug
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rng = np.random.default_rng(0)
x = f32_to_f16(rng.normal(0, 1, (B, 64, 64, 64)).astype(np.float32))
w1 = (rng.normal(0, 1, (256, 3, 3, 64)) / 24.0).astype(np.float32)
w2 = (rng.normal(0, 1, (64, 256)) / 16.0).astype(np.float32)
one, zero = np.ones(256, np.float32), np.zeros(256, np.float32)
_, ms = fused_mb_debug(x, w1, one, zero, w2, one[:64], zero[:64], x, 1, iters=5, f16=True, stamps=True)
print(f"fused_mb 64->256->64 @64, {B} frames, stamped: {ms * 1e3:.1f} us per launch", flush=True)
_, ms = fused_mb_debug(x, w1, one, zero, w2, one[:64], zero[:64], x, 1, iters=5, f16=True)
print(f"unstamped: {ms * 1e3:.1f} us per launch", flush=True)
