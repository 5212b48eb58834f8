"""Do two (three) AR batches in flight beat one? N ArEngines of 1024 windows each on N streams, started together every round.
    python tools/exp_ar_inflight.py [rounds]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isbfsar_amd import synth, weights
from isbfsar_amd.engine import ArEngine

K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
L, J, way, B = 30, 122, 60, 1024
state = weights.make_ar_state(L, J, seed=1)
ss = synth.skeleton_windows(way, L, J, seed=101)
q = torch.from_numpy(synth.skeleton_windows(B, L, J, seed=1000)).cuda()
engs, streams = [], []
for _ in range(3):
    e = ArEngine(L, J, way, device=0, precision="f16", max_batch=B)
    e.load_weights(state)
    e.set_support(poses=ss)
    engs.append(e)
    streams.append(torch.cuda.Stream())


def run(n):
    for warm in (True, False):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2 if warm else K):
            for i in range(n):
                streams[i].wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(streams[i]):
                    engs[i].infer(q)
            for a in streams[:n]:
                for b in streams[:n]:
                    if a is not b:
                        a.wait_stream(b)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return n * B * K / dt


for rep in range(2):
    for n in (1, 2, 3):
        print(f"{n} in flight: {run(n):9.0f} windows/s")
