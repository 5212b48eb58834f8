"""Does hipGraph replay of the 256-frame pose step beat eager launches? (launch gaps between the ~640 dependent kernels of a pass)
usage: [ISB_HPE_LANES=1|2] PYTHONPATH=. python tools/exp_graph_batch.py [B]"""
import sys
import time

import numpy as np
import torch

from isbfsar_amd import effnetv2, synth
from isbfsar_amd.hpe_engine import HpeEngine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
e = HpeEngine(device=0, max_batch=B)
e.load_weights(effnetv2.make_state(0))
e.set_joint_map(np.load("isbfsar_amd/assets/32_to_122.npy"), None)
fr = torch.from_numpy(synth.frames(B, seed=0)).cuda()
bb = torch.from_numpy(synth.bboxes(B, seed=0)).cuda()


def timed(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


out = {}
eager = lambda: out.update(j=e.forward(fr, bb))
print(f"eager: {timed(eager):.3f} ms per {B}-frame step")
j_eager = out["j"][0].clone()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(s):
    e.forward(fr, bb)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        res = e.forward(fr, bb)
torch.cuda.synchronize()
print(f"graph replay: {timed(g.replay):.3f} ms")
print("same bits:", bool(torch.equal(res[0], j_eager)))
