"""Loop clocks of the fused MBConv front (conv_mb8.hip mbfront8_kernel, ISB_MBF8=1): per wave of the first 64 workgroups the
cycles of its sample loop, of the waits at the loop top (tiles landed + barrier) and of the loop bodies.
usage: ISB_MBF8=1 ISB_HPE_LANES=1 PYTHONPATH=. python tools/exp_mbf8.py [B]"""
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
t = out[:64 * 4 * 4].reshape(64, 4, 4).astype(np.int64)          # the last launch's stamps
ok = t[:, :, 3] > 0
tot, wait, body, its = (t[:, :, i][ok] for i in range(4))
print(f"iterations per workgroup {its.min()}-{its.max()}; per iteration (median over waves): loop {np.median(tot / its):.0f} cycles, "
      f"waiting at the top {np.median(wait / its):.0f}, body {np.median(body / its):.0f}; whole loop {np.median(tot)} cycles")
