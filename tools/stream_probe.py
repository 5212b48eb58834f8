"""Where a single-frame step spends its time: HPE only / AR only, eager launches vs hipGraph replay."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
import torch
import bench_workloads as bw

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=200)
a = ap.parse_args()
args = argparse.Namespace(batch=1, way=60, precision="bf16")
W = bw.StreamWorkload(args, 0, 1, 0)


def timed(fn, n):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def graphed(fn):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            fn()
    torch.cuda.synchronize()
    return g.replay


def hpe():
    W.hpe.forward(W.frames, W.bbox)


win = bw.pose_windows(W.ring, W.L)


def ar():
    W.ar.infer(win)


print("step   graph  %.3f ms" % timed(W.step, a.iters))
print("step   eager  %.3f ms" % timed(W._step_eager, a.iters))
print("hpe    eager  %.3f ms" % timed(hpe, a.iters))
print("hpe    graph  %.3f ms" % timed(graphed(hpe), a.iters))
print("ar     eager  %.3f ms" % timed(ar, a.iters))
print("ar     graph  %.3f ms" % timed(graphed(ar), a.iters))
