"""Phase clocks of the all-classes attention pass (ar_kernels.hip ar_proto_all_kernel): per wave of the first 64 workgroups the cycles of
its prologue (Kq fragments, first tiles, first barrier), of its tile loop (8 classes x 14 tiles) and, inside that, of the classes'
distance epilogues (query V rows), next to the matrix pipe's share of the loop (16 MFMAs of 32 cycles per tile and wave, two waves
per SIMD). The first 64 workgroups start on a cold chip: their clocks overstate the prologue (see EXPERIMENTS.md round 4).
usage: PYTHONPATH=. python tools/exp_ar_stamps.py [B] [precision]"""
import ctypes as C
import sys
import time

import numpy as np
import torch

from isbfsar_amd import _lib, synth, weights
from isbfsar_amd.engine import ArEngine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
prec = sys.argv[2] if len(sys.argv) > 2 else "f16"
L, J, way = 30, 122, 60
e = ArEngine(L, J, way, device=0, precision=prec, max_batch=B)
e.load_weights(weights.make_ar_state(L, J, seed=1))
e.set_support(poses=synth.skeleton_windows(way, L, J, seed=101))
q = torch.from_numpy(synth.skeleton_windows(B, L, J, seed=1000)).cuda()
for _ in range(3):
    e.infer(q)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    e.infer(q)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 5 * 1e3
_lib.check(_lib.lib().isb_debug_ar_stamps(e._h, 1, None), "stamps on")
e.infer(q)
torch.cuda.synchronize()
out = np.zeros((64 * 8 * 4,), np.uint64)
_lib.check(_lib.lib().isb_debug_ar_stamps(e._h, 0, out.ctypes.data_as(C.c_void_p)), "stamps off")
t = out.reshape(64, 8, 4).astype(np.int64)
ok = t[:, :, 1] > 0
pro, loop, epi, tiles = (t[:, :, i][ok] for i in range(4))
NT = (L * (L - 1) // 2 + 31) // 32
print(f"{B} windows x {way} classes, {prec}: {ms:.3f} ms per infer() = {B / ms:.1f} k windows/s")
print(f"workgroups stamped {int(ok.any(axis=1).sum())}; per wave (median shader cycles, s_memtime): prologue {np.median(pro):.0f}, "
      f"tile loop {np.median(loop):.0f} ({np.median((loop - epi) / tiles):.0f} per tile over {int(np.median(tiles))} tiles without the "
      f"class epilogues), class epilogues {np.median(epi):.0f} ({np.median(epi / (tiles / NT)):.0f} per class)")
print("matrix work per tile and SIMD: 2 waves x 16 MFMAs x 32 cycles = 1024 cycles")
