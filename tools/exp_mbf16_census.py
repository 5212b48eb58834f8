"""When do the workgroups of mbfront16_kernel (224 -> 1344) run? Start / end of every workgroup's loop on the 100-MHz clock.
usage: ISB_EXP=0x10000 ISB_STAMP16=1 ISB_HPE_LANES=1 PYTHONPATH=. python tools/exp_mbf16_census.py [B]"""
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
t = out[:2000].reshape(1000, 2).astype(np.int64)
d = (t[:, 1] - t[:, 0]) / 100.0
n = int((t[:, 1] > 0).sum())
print(f"{n} workgroups wrote stamps")
for lo, hi in ((0, 32), (32, 128), (128, 256), (256, 384), (384, 539), (539, 650), (650, 759)):
    x = d[lo:hi]
    x = x[(x > 0) & (x < 1000)]
    print(f"workgroups {lo:3d}-{hi:3d}: loop duration median {np.median(x):6.1f} us, min {x.min():6.1f}, max {x.max():6.1f} ({len(x)} valid)")
for sl in range(11):
    x = d[sl:759:11]
    x = x[(x > 0) & (x < 1000)]
    print(f"slice {sl:2d}: median {np.median(x):6.1f} us")
