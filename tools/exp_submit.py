"""Host batches: the synchronous entry, two batches in flight (submit / wait), the resident rate and the bare 236-MB copy."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from isbfsar_amd.hpe_engine import HpeEngine
from isbfsar_amd import effnetv2
from isbfsar_amd import synth
a = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "isbfsar_amd", "assets")
B = 256
e = HpeEngine(device=0, max_batch=B)
e.set_joint_map(np.load(os.path.join(a, "32_to_122.npy")), None)
e.load_weights(effnetv2.make_state(0))
fr = synth.frames(B, seed=1); bb = synth.bboxes(B, seed=1)
p = [torch.from_numpy(fr).pin_memory().numpy() for _ in range(2)]
dev_f = torch.from_numpy(fr).cuda(); dev_b = torch.from_numpy(bb).cuda()
def t(f, n=8):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print("resident            %.2f ms" % t(lambda: e.forward(dev_f, dev_b)))
print("host sync (ROI)     %.2f ms" % t(lambda: e.forward(p[0], bb)))
k = [0]
def pipe():
    e.submit(p[k[0] & 1], bb); k[0] += 1
    if k[0] >= 2: e.wait()
print("submit(k+1) + wait(k)  %.2f ms" % t(pipe, 10))
# the copy alone
s = torch.cuda.Stream()
d = torch.empty_like(dev_f)
src = torch.from_numpy(p[0])
def cp():
    with torch.cuda.stream(s): d.copy_(src, non_blocking=True)
    s.synchronize()
print("236 MB H2D alone    %.2f ms" % t(cp))
