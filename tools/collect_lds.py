#!/usr/bin/env python3
"""rocprofv3 --pmc pass with the SQ LDS counters (SQ_INSTS_LDS, SQ_ACTIVE_INST_LDS, SQ_LDS_BANK_CONFLICT, SQ_LDS_IDX_ACTIVE,
SQ_WAIT_INST_LDS, SQ_BUSY_CYCLES, SQ_WAVE_CYCLES) -> per kernel: LDS instructions per dispatch, bank-conflict cycles as a share of
the cycles the LDS index unit was active, and the share of wave cycles spent waiting on LDS.
    python tools/collect_lds.py out.json <counter_collection.csv>
Whatever subset of the counters the pass could collect is reported (a counter this ROCm does not expose is simply absent)."""
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
for k, d in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", kv[1].get("SQ_BUSY_CYCLES", 0.0))):
    e = {"dispatches": max(len(v) for (kk, c), v in disp.items() if kk == k)}
    for c, v in d.items():
        e[c] = v / max(1, len(disp[(k, c)]))
    if e.get("SQ_LDS_IDX_ACTIVE"):
        e["bank_conflict_share_of_lds_active"] = round(e.get("SQ_LDS_BANK_CONFLICT", 0.0) / e["SQ_LDS_IDX_ACTIVE"], 4)
    if e.get("SQ_WAVE_CYCLES") and "SQ_WAIT_INST_LDS" in e:
        e["wave_cycles_waiting_on_lds"] = round(e["SQ_WAIT_INST_LDS"] / e["SQ_WAVE_CYCLES"], 4)
    out["kernels"][k] = e
json.dump(out, open(sys.argv[1], "w"), indent=1)
for k, e in list(out["kernels"].items())[:12]:
    print(f"  {k[:70]:70s} n={e['dispatches']:4d} conflicts/active={e.get('bank_conflict_share_of_lds_active')} wait_lds={e.get('wave_cycles_waiting_on_lds')}")
