"""Per-layer-group time of conv_igemm launches of the LAST backbone pass in a rocprofv3 kernel trace.
usage: python tools/layer_breakdown.py <kernel_trace.csv> [B]"""
import collections
import csv
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isbfsar_amd import effnetv2 as E

rows = list(csv.DictReader(open(sys.argv[1])))
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
ig = [r for r in rows if 'conv_igemm' in r['Kernel_Name'] or 'gemm1x1' in r['Kernel_Name'] or 'isb::conv3x3_dma' in r['Kernel_Name']]
convs = []
for b in E.blocks():
    if b.kind == 'fused':
        if b.cexp != b.cin:
            convs.append(('exp3x3', b.in_hw, b.out_hw, b.cin, b.cexp, 3, b.stride))
            convs.append(('proj', b.out_hw, b.out_hw, b.cexp, b.cout, 1, 1))
        else:
            convs.append(('f3x3', b.in_hw, b.out_hw, b.cin, b.cout, 3, b.stride))
    else:
        convs.append(('exp1x1', b.in_hw, b.in_hw, b.cin, b.cexp, 1, 1))
        convs.append(('proj', b.out_hw, b.out_hw, b.cexp, b.cout, 1, 1))
convs.append(('head', 8, 8, 640, 1280, 1, 1))
n = len(convs)
last = ig[-n:]
agg = collections.OrderedDict()
tot = 0
for c, r in zip(convs, last):
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
    fl = 2.0 * B * c[2] * c[2] * c[5] * c[5] * c[3] * c[4]
    by = B * (c[1] * c[1] * c[3] + c[2] * c[2] * c[4]) * 2
    key = (c[0], c[1], c[3], c[4], c[6])
    a = agg.setdefault(key, [0, 0, 0, 0, r['Kernel_Name'][:60]])
    a[0] += d; a[1] += fl; a[2] += by; a[3] += 1
    tot += d
print('launches', len(ig), 'per pass', n, 'last-pass total ms', round(tot, 3))
for k, a in agg.items():
    print(f"{str(k):38s} n={a[3]:2d} ms={a[0]:6.3f} ({100*a[0]/tot:4.1f}%) TF={a[1]/a[0]/1e9:6.0f} GB/s={a[2]/a[0]/1e6:6.0f} {a[4]}")
