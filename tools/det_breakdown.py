"""Per-layer-shape time and TFLOP/s of the detector's convolutions in the LAST pass of a rocprofv3 kernel trace:
python tools/det_breakdown.py <kernel_trace.csv> [B]"""
import collections
import csv
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isbfsar_amd import _lib

h = _lib.lib()
L = []
for i in range(h.isb_det_n_convs()):
    name = C.create_string_buffer(128)
    dims = (C.c_int32 * 8)()
    h.isb_det_describe(i, name, 128, dims)
    L.append((name.value.decode(), list(dims)))


def in_hw(name):          # input resolution of a layer at 256 x 256 (module order of the public YOLOv4)
    p, leaf = name.split(".", 1)
    if p == "down1":
        return 256 if leaf in ("conv1", "conv2") else 128
    if p in ("down2", "down3", "down4", "down5"):
        o = {"down2": 64, "down3": 32, "down4": 16, "down5": 8}[p]
        return 2 * o if leaf == "conv1" else o
    n = int(leaf.replace("conv", ""))
    if p == "neek":
        return 8 if n <= 7 else (16 if n <= 14 and n != 14 or n == 8 else 32) if n != 15 else 32
    return 32 if n <= 3 else (16 if n <= 11 else 8)


rows = [r for r in csv.DictReader(open(sys.argv[1])) if any(k in r["Kernel_Name"] for k in ("conv_igemm", "gemm1x1", "conv3x3_dma", "det_stem"))]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
last = rows[-len(L):]
agg = collections.OrderedDict()
tot = 0.0
for (name, d), r in zip(L, last):
    cin, cout, k, stride = d[:4]
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    hw = in_hw(name)
    if name == "neek.conv8":
        hw = 16
    if name == "neek.conv15":
        hw = 32
    ohw = hw // stride
    fl = 2.0 * B * ohw * ohw * k * k * cin * cout
    key = f"{k}x{k} s{stride} {cin}->{cout} @{ohw}"
    a = agg.setdefault(key, [0.0, 0.0, 0, r["Kernel_Name"].replace("void isb::", "")[:36]])
    a[0] += us; a[1] += fl; a[2] += 1
    tot += us
print(f"detector convolutions per pass {len(L)}; last pass {tot / 1e3:.3f} ms ({sum(a[1] for a in agg.values()) / tot / 1e6:.0f} TFLOP/s), B={B}")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"{k:30s} n={a[2]:2d} us={a[0]:7.1f} ({100 * a[0] / tot:4.1f}%) TFLOP/s={a[1] / a[0] / 1e6:6.0f}  {a[3]}")
