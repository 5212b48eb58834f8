#!/usr/bin/env python3
"""rocprofv3 --pmc pass(es) with SQ_VALU_MFMA_BUSY_CYCLES / SQ_INSTS_MFMA / SQ_BUSY_CYCLES / SQ_ACTIVE_INST_VALU / GRBM_GUI_ACTIVE
-> per kernel: matrix-pipe busy share of the dispatch.
    python tools/collect_mfma.py out.json <counter_collection.csv> [more csvs ...]
MFMA busy share = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): SQ_VALU_MFMA_BUSY_CYCLES counts SIMD cycles
(32 per v_mfma_f32_32x32x16_bf16, summed over the chip), GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md).
Under counter collection dispatches run serialised and slower than in a plain run, so the share is a LOWER bound on what
the same kernel reaches un-profiled; the instruction counts are exact."""
import collections
import csv
import json
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if "isb::" not in k:
            continue
        k = k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
out = {"_how": __doc__, "kernels": {}}
for k, d in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)):
    n = max(len(v) for (kk, c), v in disp.items() if kk == k)
    e = {"dispatches": n}
    for c, v in d.items():
        e[c] = v / max(1, len(disp[(k, c)]))          # per dispatch
    if e.get("GRBM_GUI_ACTIVE"):
        simd_cycles = 1024.0 * e["GRBM_GUI_ACTIVE"] / 8.0
        e["mfma_busy_share"] = round(e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / simd_cycles, 4)
        if "SQ_ACTIVE_INST_VALU" in e:
            e["valu_active_share"] = round(4.0 * e["SQ_ACTIVE_INST_VALU"] / simd_cycles, 4)     # quad-cycles -> cycles
    out["kernels"][k] = e
json.dump(out, open(sys.argv[1], "w"), indent=1)
tot_busy = sum(e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) * e["dispatches"] for e in out["kernels"].values())
tot_cyc = sum(1024.0 * e.get("GRBM_GUI_ACTIVE", 0) / 8.0 * e["dispatches"] for e in out["kernels"].values())
print(f"{len(out['kernels'])} kernels; matrix-pipe busy share over all dispatches {tot_busy / max(tot_cyc, 1):.3f}")
for k, e in list(out["kernels"].items())[:14]:
    print(f"  {k[:70]:70s} n={e['dispatches']:4d} mfma_busy={e.get('mfma_busy_share')} valu_active={e.get('valu_active_share')}")
