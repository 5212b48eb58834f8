#!/usr/bin/env python3
"""Per-kernel A/B of two rocprofv3 --kernel-trace --stats summaries taken in ONE box session:
    python tools/kstat_diff.py A_kernel_stats.csv B_kernel_stats.csv [min_total_ms]
prints, for every kernel above min_total_ms in either run, calls, average microseconds in A and B and B / A."""
import csv
import re
import sys


def load(p):
    out = {}
    for r in csv.DictReader(open(p)):
        name = re.sub(r"\(.*", "", r["Name"].replace("(anonymous namespace)::", "")).replace("isb::", "")
        out[name] = (int(r["Calls"]), float(r["TotalDurationNs"]))
    return out


a, b = load(sys.argv[1]), load(sys.argv[2])
floor = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
rows = []
for k in sorted(set(a) | set(b)):
    ca, ta = a.get(k, (0, 0.0))
    cb, tb = b.get(k, (0, 0.0))
    if max(ta, tb) / 1e6 < floor:
        continue
    rows.append((max(ta, tb), k, ca, ta / max(ca, 1) / 1e3, cb, tb / max(cb, 1) / 1e3))
ta_all = sum(v[1] for v in a.values()) / 1e6
tb_all = sum(v[1] for v in b.values()) / 1e6
print(f"all kernels: A {ta_all:.2f} ms, B {tb_all:.2f} ms, B/A {tb_all / ta_all:.4f}")
for _, k, ca, ua, cb, ub in sorted(rows, reverse=True):
    print(f"{k[:70]:70s} calls {ca:5d}/{cb:5d}  A {ua:9.2f} us  B {ub:9.2f} us  B/A {ub / ua if ua else float('nan'):.3f}")
