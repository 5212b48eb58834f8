"""Do the two lanes of a 256-frame pose step lose by starting each step TOGETHER? isb_hpe_forward forks lane 1 at the start of a call and
joins it at the end, so both lanes walk the same stages at the same time. Here: two one-lane engines of 128 frames on two streams,
(a) joined after every step (what the library does), (b) free-running, (c) free-running with lane B started half a pass late.
    python tools/exp_lane_offset.py [steps] [frames per engine: 128 | 256 = two whole 256-frame steps in flight]"""
import os
import sys
import time

os.environ["ISB_HPE_LANES"] = "1"
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isbfsar_amd import effnetv2, synth
from isbfsar_amd.hpe_engine import HpeEngine

K = int(sys.argv[1]) if len(sys.argv) > 1 else 10
N = int(sys.argv[2]) if len(sys.argv) > 2 else 128          # frames per engine and step
assets = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "isbfsar_amd", "assets")
state = effnetv2.make_state(0)
engs = []
for _ in range(2):
    e = HpeEngine(device=0, max_batch=N)
    e.load_weights(state)
    e.set_joint_map(np.load(os.path.join(assets, "32_to_122.npy")), None)
    engs.append(e)
fr = torch.from_numpy(synth.frames(2 * N, seed=1)).cuda()
bb = torch.from_numpy(synth.bboxes(2 * N, seed=1)).cuda()
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()


def run(mode, offset_frames=0):
    torch.cuda.synchronize()
    for warm in (True, False):
        if offset_frames and not warm:
            with torch.cuda.stream(sB):
                engs[1].forward(fr[:offset_frames], bb[:offset_frames])
        torch.cuda.synchronize() if warm else None
        t0 = time.perf_counter()
        for _ in range(3 if warm else K):
            with torch.cuda.stream(sA):
                engs[0].forward(fr[:N], bb[:N])
            with torch.cuda.stream(sB):
                engs[1].forward(fr[N:], bb[N:])
            if mode == "joined":
                sA.wait_stream(sB)
                sB.wait_stream(sA)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return 1e3 * dt / K * 128 / N


for rep in range(3):
    print("joined  %.3f ms / 256 frames" % run("joined"))
    print("free    %.3f" % run("free"))
    print("free+64 %.3f (includes the 64-frame head start: %d steps)" % (run("free", 64), K))
    print("free+32 %.3f" % run("free", 32))
for e in engs:
    e.close()
