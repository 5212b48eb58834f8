#!/usr/bin/env python3
"""Copies the evidence that tools/round_profiles.sh left under gpurun_out/ into profiles/ (tracked), named per round:
    python tools/copy_profiles.py r01
Counter CSVs are trimmed to dispatch / kernel / grid / workgroup / counter columns."""
import csv
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
G, P = "gpurun_out", "profiles"
for w, name in (("pipeline", "pipe"), ("hpe", "hpe"), ("ar", "ar"), ("stream", "stream"), ("det", "det")):
    shutil.copy(f"{G}/bench_{w}.json", f"{P}/{tag}_bench_{name}.json")
shutil.copy(f"{G}/prof_pipe/run_kernel_stats.csv", f"{P}/{tag}_pipeline_b256_kernel_stats.csv")
shutil.copy(f"{G}/prof_ar/run_kernel_stats.csv", f"{P}/{tag}_ar_b1024_kernel_stats.csv")
shutil.copy(f"{G}/prof_hpe1/run_kernel_stats.csv", f"{P}/{tag}_hpe_b256_onelane_kernel_stats.csv")
shutil.copy(f"{G}/traffic.json", f"{P}/{tag}_traffic.json")
import os
for src, dst in (("bench_ar_bf16x3.json", "bench_ar_bf16x3.json"), ("bench_hpe_host.json", "bench_hpe_host_input.json"),
                 ("bench_pipe_b2048.json", "bench_pipe_b2048.json"), ("mfma_hpe.json", "mfma_counters_hpe_b256_onelane.json"),
                 ("mfma_ar.json", "mfma_counters_ar_b1024.json"), ("layer_breakdown.txt", "hpe_b256_layer_breakdown.txt"),
                 ("dw_breakdown.txt", "hpe_b256_depthwise_breakdown.txt"), ("det_breakdown.txt", "det_b256_layer_breakdown.txt"),
                 ("bench_hpe_host_wholeframes.json", "bench_hpe_host_input_wholeframes.json"),
                 ("bench_hpe_host_pipelined.json", "bench_hpe_host_input_pipelined.json"),
                 ("bench_hpe_bf16_everywhere.json", "bench_hpe_bf16_everywhere.json"), ("estimate_latency.json", "estimate_latency.json"),
                 ("bench_hpe_bf16_f16tail.json", "bench_hpe_bf16_f16tail.json"), ("bench_ar_bf16.json", "bench_ar_bf16.json"),
                 ("lds_hpe.json", "lds_counters_hpe_b256_onelane.json"),
                 ("prof_det/run_kernel_stats.csv", "det_b256_kernel_stats.csv")):
    if os.path.exists(f"{G}/{src}"):
        shutil.copy(f"{G}/{src}", f"{P}/{tag}_{dst}")
for src, dst in (("pmc_fetch", "hpe_b256_fetch_size"), ("pmc_write", "hpe_b256_write_size")):
    with open(f"{G}/{src}/run_counter_collection.csv") as f, open(f"{P}/{tag}_{dst}_counters.csv", "w", newline="") as o:
        w = csv.writer(o)
        w.writerow(["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "Counter_Name", "Counter_Value"])
        for r in csv.DictReader(f):
            w.writerow([r["Dispatch_Id"], r["Kernel_Name"][:80], r["Grid_Size"], r["Workgroup_Size"], r["Counter_Name"],
                        r["Counter_Value"]])
for src, dst in (("pmc_l2/l2_table.txt", "l2_counters_hpe_b256_onelane.txt"), ("pmc_l2/summary.txt", "l2_counters_hpe_b256_onelane_raw.txt"),
                 ("fronts_three_forms.txt", "fused_fronts_three_forms.txt"), ("pmc_fronts/summary.txt", "fused_fronts_sq_lds_counters.txt"),
                 ("ab_mbf16_forms.txt", "ab_kernels_mbfront16_form1_vs_form2.txt"), ("ab_mbf8_forms.txt", "ab_kernels_mbfront8_form1_vs_form2.txt")):
    if os.path.exists(f"{G}/{src}"):
        shutil.copy(f"{G}/{src}", f"{P}/{tag}_{dst}")
print("copied to", P)
