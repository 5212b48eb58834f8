"""Single-frame pose latency three ways: HpeEngine.forward on a pageable numpy frame (what HumanPoseEstimator.estimate() calls), the
same pass on device tensors issued eagerly, and a hipGraph replay of it (torch.cuda.CUDAGraph over the library's launches).
usage: PYTHONPATH=. python tools/exp_b1_graph.py"""
import time

import numpy as np
import torch

from isbfsar_amd import effnetv2, synth
from isbfsar_amd.hpe_engine import HpeEngine

e = HpeEngine(device=0, max_batch=1)
e.load_weights(effnetv2.make_state(0))
e.set_joint_map(np.load("isbfsar_amd/assets/32_to_122.npy"), None)
fh = synth.frames(1, seed=0)
bh = synth.bboxes(1, seed=0)
fd = torch.from_numpy(fh).cuda()
bd = torch.from_numpy(bh).cuda()


def med(fn, n=200):
    for _ in range(10):
        fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3


def eager_dev():
    e.forward(fd, bd)
    torch.cuda.synchronize()


print(f"numpy in, numpy out (isb_hpe_forward_host): {med(lambda: e.forward(fh, bh)):.3f} ms")
print(f"device tensors, eager + synchronize:        {med(eager_dev):.3f} ms")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3):
        out = e.forward(fd, bd)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        out = e.forward(fd, bd)


def replay():
    g.replay()
    torch.cuda.synchronize()


print(f"device tensors, hipGraph replay + synchronize: {med(replay):.3f} ms")
pin = torch.from_numpy(fh).pin_memory()
res = torch.empty((1, 122, 3), dtype=torch.float32).pin_memory()


def replay_host():
    fd.copy_(pin, non_blocking=True)
    g.replay()
    res.copy_(out[0], non_blocking=True)
    torch.cuda.synchronize()


with torch.cuda.stream(s):
    print(f"pinned frame H2D + replay + pose D2H + synchronize: {med(replay_host):.3f} ms")
