"""Phase clocks of the fused 8 x 8 MBConv chain (conv_mb8.hip): s_memtime stamps of wave 0 / the last wave of the first 32 workgroups
for one 384 -> 2304 -> 384 block and one 640 -> 3840 -> 640 block, plus the chain's share of a 256-frame pose step.
usage: python tools/exp_mb8.py [B] [precision]"""
import ctypes as C
import sys
import time

import numpy as np
import torch

from isbfsar_amd import _lib, effnetv2, synth
from isbfsar_amd.hpe_engine import HpeEngine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
prec = sys.argv[2] if len(sys.argv) > 2 else "f16"
e = HpeEngine(device=0, max_batch=B, precision=prec)
e.load_weights(effnetv2.make_state(0))
e.set_joint_map(np.load("isbfsar_amd/assets/32_to_122.npy"), None)
fr = torch.from_numpy(synth.frames(B, seed=0)).cuda()
bb = torch.from_numpy(synth.bboxes(B, seed=0)).cuda()
for _ in range(3):
    e.forward(fr, bb)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    e.forward(fr, bb)
torch.cuda.synchronize()
print(f"pose step B={B} {prec}: {(time.perf_counter() - t0) * 100:.3f} ms")
_lib.check(_lib.lib().isb_debug_hpe_mb8_stamps(e._h, 1, None), "stamps on")
e.forward(fr, bb)
torch.cuda.synchronize()
out = np.zeros((32, 2, 32), np.uint64)
_lib.check(_lib.lib().isb_debug_hpe_mb8_stamps(e._h, 0, out.ctypes.data_as(C.c_void_p)), "stamps off")
names = ["phases 1-3 (expand, depthwise, pool)", "barrier", "FC1 partials", "barrier + hidden units", "FC2 + sigmoid", "barrier",
         "projection k loop", "epilogue", "barrier"]
for blk, label in ((0, "384 -> 2304 -> 384"), (1, "640 -> 3840 -> 640")):
    for w0, wl in ((0, "wave 0"), (16, "last wave (a helper in phase 5)")):
        t = out[:, blk, w0:w0 + 10].astype(np.int64)
        ok = t[:, 0] > 0
        if not ok.any():
            continue
        d = np.diff(t[ok], axis=1)
        print(f"block {label}, {wl}: total {np.median(t[ok, 9] - t[ok, 0])} cycles (100 MHz s_memtime ticks x 1 = shader clocks? see DESIGN)")
        for i, n in enumerate(names):
            print(f"    {n:40s} median {int(np.median(d[:, i])):8d}   min {int(d[:, i].min()):8d}   max {int(d[:, i].max()):8d}")
