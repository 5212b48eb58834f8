"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md prescribes) of
`ISB_HPE_LANES=1 python3 bench.py --workload hpe --steps 1 --warmup 1 --no-cpu-baseline` into profiles/rNN_traffic.json:
HBM bytes per forward pass (B=256 frames) for each kernel family.
usage: python tools/collect_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>
FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is doubled (gfx950 tallies the 128-B requests of 16-B/lane reads at 64 B)."""
import collections
import csv
import json
import sys


def family(name: str) -> str:
    if "dwconv3x3" in name:
        return "dwconv3x3_pool"
    if "conv_igemm" in name or "gemm1x1" in name or "conv3x3_dma" in name or "conv3x3_c32_rows" in name or "fused_mb" in name or "mbfront8" in name or "mbfront16" in name or "mb8_chain" in name:
        return "conv_igemm"
    if "se_fc" in name:
        return "se_fc"
    if "ar_" in name:
        return "ar"
    return "other" if "isb::" in name else "_skip"


def load(path, counter):
    tot = collections.Counter()
    n = collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        f = family(r["Kernel_Name"])
        if f == "_skip":
            continue
        tot[f] += float(r["Counter_Value"])
        n[f] += 1
    return tot, n


fetch, nf = load(sys.argv[1], "FETCH_SIZE")
write, nw = load(sys.argv[2], "WRITE_SIZE")
forwards = sum(1 for r in csv.DictReader(open(sys.argv[1]))     # one stem launch per forward pass (one lane)
               if r["Counter_Name"] == "FETCH_SIZE" and "stem_kernel" in r["Kernel_Name"])
out = {"_how": __doc__, "forwards_in_trace": forwards, "hpe_b256": {}}
for f in sorted(set(fetch) | set(write)):
    fk, wk = fetch[f] / forwards, write[f] / forwards
    out["hpe_b256"][f] = {"FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk, "launches": nf[f] / forwards,
                          "hbm_bytes_per_forward": (2.0 * fk + wk) * 1024.0}
# optional: the same two passes of `bench.py --workload ar --steps 1 --warmup 1 --no-cpu-baseline` (argv[4], argv[5])
if len(sys.argv) > 5:
    import re

    # the all-classes pass, whatever its operand type (the arg-max class's pass is ar_proto_chosen_kernel / ar_stats_kernel<true, ..>:
    # bf16 hi + lo images)
    ALL_CLASSES = {"ar_proto": re.compile(r"ar_proto_all_kernel<"), "ar_stats": re.compile(r"ar_stats_kernel<\s*false\b")}

    def per_launch(path, counter, rx):
        tot = n = 0
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter and rx.search(r["Kernel_Name"]):
                tot += float(r["Counter_Value"])
                n += 1
        return tot / max(n, 1), n
    out["ar_b1024"] = {}
    for name, rx in ALL_CLASSES.items():
        fk, n = per_launch(sys.argv[4], "FETCH_SIZE", rx)
        wk, _ = per_launch(sys.argv[5], "WRITE_SIZE", rx)
        if n == 0:
            raise SystemExit(f"collect_traffic: no {name} launch matched in {sys.argv[4]} -- refusing to write a zero")
        out["ar_b1024"][name] = {"FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk, "launches_in_trace": n,
                                 "hbm_bytes_per_launch": (2.0 * fk + wk) * 1024.0}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: round(v["hbm_bytes_per_forward"] / 1e9, 2) for k, v in out["hpe_b256"].items()}), "GB per forward;", forwards, "forwards")
