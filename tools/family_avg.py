#!/usr/bin/env python3
"""Average launch duration of the convolution family in a rocprofv3 --stats summary, next to the figure bench.py's
roofline object reports (HIP events around the same launches):
    python tools/family_avg.py <kernel_stats.csv> [bench line json]"""
import csv
import json
import sys

FAMILY = ("conv_igemm", "gemm1x1", "conv3x3_dma", "conv3x3_c32_rows", "fused_mb", "mbfront8", "mbfront16", "splitk_reduce")
rows = [r for r in csv.DictReader(open(sys.argv[1])) if any(k in r["Name"] for k in FAMILY)]
calls = sum(int(r["Calls"]) for r in rows)
total_ns = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"rocprofv3: {calls} launches of the convolution family, {total_ns / calls / 1e6:.5f} ms average, "
      f"{total_ns / 1e6:.2f} ms in total")
if len(sys.argv) > 2:
    d = json.load(open(sys.argv[2]))["roofline"]
    print(f"bench.py : {d['launches']} launches timed with HIP events, {d['avg_launch_ms']:.5f} ms average "
          f"({d['achieved']} {d['unit']})")
