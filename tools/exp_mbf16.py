"""Phase clocks of the fused MBConv front of the 16 x 16 stage (conv_mb16.hip mbfront16_kernel, 224 -> 1344 blocks): per wave of the
first 32 workgroups the cycles of its loop and of the phases of a band step.
usage: ISB_STAMP16=1 ISB_HPE_LANES=1 PYTHONPATH=. python tools/exp_mbf16.py [B]"""
import ctypes as C
import sys

import numpy as np
import torch

from isbfsar_amd import _lib, effnetv2, synth
from isbfsar_amd.hpe_engine import HpeEngine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
e = HpeEngine(device=0, max_batch=B)
e.load_weights(effnetv2.make_state(0))
e.set_joint_map(np.load("isbfsar_amd/assets/32_to_122.npy"), None)
fr = torch.from_numpy(synth.frames(B, seed=0)).cuda()
bb = torch.from_numpy(synth.bboxes(B, seed=0)).cuda()
for _ in range(3):
    e.forward(fr, bb)
torch.cuda.synchronize()
_lib.check(_lib.lib().isb_debug_hpe_mb8_stamps(e._h, 1, None), "stamps on")
e.forward(fr, bb)
torch.cuda.synchronize()
out = np.zeros((32 * 2 * 32,), np.uint64)
_lib.check(_lib.lib().isb_debug_hpe_mb8_stamps(e._h, 0, out.ctypes.data_as(C.c_void_p)), "stamps off")
t = out[:32 * 6 * 10].reshape(32, 6, 10).astype(np.int64)        # the last launch's stamps (the last 224 -> 1344 block)
ok = t[:, :, 1] > 0                                              # (four of the six wave slots are written)
n = t[:, :, 1][ok]
names = ["tile wait + barrier", "expand MFMAs", "second barrier", "E epilogue (+ next tile's requests)", "depthwise MFMAs", "SiLU + D rows", "pooled means (per sample)"]
print(f"band steps per wave {n.min()}-{n.max()} (9 per sample); loop {np.median(t[:, :, 0][ok])} cycles = {np.median(t[:, :, 0][ok] / n):.0f} per step")
for i, nm in enumerate(names):
    print(f"  {nm:38s} {np.median(t[:, :, 2 + i][ok] / n):7.0f} cycles per step")
rt = t[:, :, 9][ok]
print(f"loop: {np.median(rt) / 100:.1f} us by the 100-MHz clock -> the chip held {np.median(t[:, :, 0][ok] / rt) * 100:.0f} MHz in it")
