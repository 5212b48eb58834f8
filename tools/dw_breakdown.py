"""Per-shape time and HBM rate of the depthwise launches of the LAST backbone pass in a rocprofv3 kernel trace (one lane):
python tools/dw_breakdown.py <kernel_trace.csv> [B]   (bytes = input + output of the launch, bf16 / fp16)"""
import collections
import csv
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isbfsar_amd import effnetv2 as E

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "dwconv3x3" in r["Kernel_Name"]]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
seq = [(f"dw {b.cexp}ch {b.in_hw}->{b.out_hw}", 2.0 * B * b.cexp * (b.in_hw ** 2 + b.out_hw ** 2)) for b in E.blocks() if b.kind == "mb"
       and not (b.stride == 1 and B >= 32 and ((b.in_hw == 8 and b.cin == 384) or (b.in_hw == 16 and b.cin in (192, 224))))]       # those run inside mbfront8 / mbfront16 (layer_breakdown.py)
last = rows[-len(seq):]
agg = collections.OrderedDict()
tot = 0.0
for (label, by), r in zip(seq, last):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    a = agg.setdefault(label, [0.0, 0.0, 0, r["Kernel_Name"].replace("void isb::", "")[:40]])
    a[0] += d; a[1] += by; a[2] += 1
    tot += d
print(f"depthwise launches per pass {len(seq)}; last pass {tot:.3f} ms ({sum(b for _, b in seq) / tot / 1e9:.2f} TB/s), B={B}")
for k, a in agg.items():
    print(f"{k:24s} n={a[2]:2d} ms={a[0]:6.3f} avg {1e3 * a[0] / a[2]:6.1f} us  {a[1] / a[0] / 1e9:5.2f} TB/s  {a[3]}")
