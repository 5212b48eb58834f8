#!/bin/bash
# Regenerates the round's evidence under gpurun_out/ (copy into profiles/ afterwards: tools/copy_profiles.py rNN). Run on the MI355X box:
#   bash tools/round_profiles.sh rNN
ROUND=${1:?usage: round_profiles.sh rNN}
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== PMC traffic"
ISB_BENCH_INFLIGHT=1 ISB_HPE_LANES=1 timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o run -- python3 bench.py --workload hpe --steps 1 --warmup 1 --no-cpu-baseline --no-extras --min-gpu-seconds 0 > gpurun_out/pmc_fetch.log 2>&1
ISB_BENCH_INFLIGHT=1 ISB_HPE_LANES=1 timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o run -- python3 bench.py --workload hpe --steps 1 --warmup 1 --no-cpu-baseline --no-extras --min-gpu-seconds 0 > gpurun_out/pmc_write.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_ar_fetch -o run -- python3 bench.py --workload ar --steps 1 --warmup 1 --no-cpu-baseline --no-extras --min-gpu-seconds 0 > gpurun_out/pmc_ar_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_ar_write -o run -- python3 bench.py --workload ar --steps 1 --warmup 1 --no-cpu-baseline --no-extras --min-gpu-seconds 0 > gpurun_out/pmc_ar_write.log 2>&1
python3 tools/collect_traffic.py gpurun_out/pmc_fetch/run_counter_collection.csv gpurun_out/pmc_write/run_counter_collection.csv gpurun_out/traffic.json gpurun_out/pmc_ar_fetch/run_counter_collection.csv gpurun_out/pmc_ar_write/run_counter_collection.csv
cp gpurun_out/traffic.json profiles/${ROUND}_traffic.json   # bench.py reads the latest profiles/rNN_traffic.json: the lines below carry THIS session's bytes
echo "== bench lines"
for w in pipeline hpe ar stream det; do
  timeout -k 10 400 python bench.py --workload $w $( [ $w = stream ] && echo "--steps 300 --warmup 20" ) $( [ $w != pipeline ] && echo "--min-gpu-seconds 0" ) > gpurun_out/bench_$w.log 2>&1
  tail -1 gpurun_out/bench_$w.log > gpurun_out/bench_$w.json
  python3 -c "import json;d=json.load(open('gpurun_out/bench_$w.json'));print('$w',d['value'],d['unit'],d['ms_per_step'],'ms',d['roofline']['achieved'],d['roofline']['frac'],d['cpu_baseline']['value'] if d.get('cpu_baseline') else None, d.get('parity'))"
done
echo "== extra lines: AR in bf16x3 (the reference arithmetic is fp32), pose stage from pinned host frames, configs[3] at its full single-GPU batch"
timeout -k 10 300 python bench.py --workload ar --precision bf16x3 --no-cpu-baseline --min-gpu-seconds 0 2>/dev/null | tail -1 > gpurun_out/bench_ar_bf16x3.json
timeout -k 10 300 python bench.py --workload hpe --host-input --no-cpu-baseline --min-gpu-seconds 0 2>/dev/null | tail -1 > gpurun_out/bench_hpe_host.json
ISB_HPE_ROI=0 timeout -k 10 300 python bench.py --workload hpe --host-input --no-cpu-baseline --min-gpu-seconds 0 2>/dev/null | tail -1 > gpurun_out/bench_hpe_host_wholeframes.json
timeout -k 10 300 python bench.py --workload hpe --host-input --pipelined --no-cpu-baseline --min-gpu-seconds 0 2>/dev/null | tail -1 > gpurun_out/bench_hpe_host_pipelined.json
timeout -k 10 300 python bench.py --workload hpe --hpe-precision bf16 --min-gpu-seconds 0 2>/dev/null | tail -1 > gpurun_out/bench_hpe_bf16_everywhere.json
timeout -k 10 300 python bench.py --workload hpe --hpe-precision bf16_f16tail --min-gpu-seconds 0 2>/dev/null | tail -1 > gpurun_out/bench_hpe_bf16_f16tail.json
timeout -k 10 300 python bench.py --workload ar --precision bf16 --no-cpu-baseline --min-gpu-seconds 0 2>/dev/null | tail -1 > gpurun_out/bench_ar_bf16.json
timeout -k 10 400 python bench.py --workload pipeline --batch 2048 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --min-gpu-seconds 0 2>/dev/null | tail -1 > gpurun_out/bench_pipe_b2048.json
timeout -k 10 300 python tools/estimate_latency.py > gpurun_out/estimate_latency.json 2>gpurun_out/estimate_latency.err || true
for f in ar_bf16x3 ar_bf16 hpe_host hpe_host_wholeframes hpe_host_pipelined hpe_bf16_everywhere hpe_bf16_f16tail pipe_b2048; do python3 -c "import json;d=json.load(open('gpurun_out/bench_$f.json'));print('$f',d['value'],d['unit'],d['ms_per_step'],'ms')"; done
echo "== kernel stats (pipeline)"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pipe -o run -- python3 bench.py --workload pipeline --steps 5 --warmup 2 --no-cpu-baseline --no-extras --min-gpu-seconds 0 > gpurun_out/prof_pipe.log 2>&1
echo "== kernel stats (hpe, one lane: every convolution launch is a 256-frame launch, as in bench.py's roofline pass)"
ISB_BENCH_INFLIGHT=1 ISB_HPE_LANES=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_hpe1 -o run -- python3 bench.py --workload hpe --steps 5 --warmup 2 --no-cpu-baseline --min-gpu-seconds 0 > gpurun_out/prof_hpe1.log 2>&1
python3 tools/family_avg.py gpurun_out/prof_hpe1/run_kernel_stats.csv gpurun_out/bench_hpe.json
python3 tools/layer_breakdown.py gpurun_out/prof_hpe1/run_kernel_trace.csv > gpurun_out/layer_breakdown.txt 2>&1 || true
python3 tools/dw_breakdown.py gpurun_out/prof_hpe1/run_kernel_trace.csv > gpurun_out/dw_breakdown.txt 2>&1 || true
echo "== kernel stats (det)"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_det -o run -- python3 bench.py --workload det --steps 5 --warmup 2 --no-cpu-baseline --min-gpu-seconds 0 > gpurun_out/prof_det.log 2>&1
python3 tools/det_breakdown.py gpurun_out/prof_det/run_kernel_trace.csv > gpurun_out/det_breakdown.txt 2>&1 || true
echo "== kernel stats (ar)"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ar -o run -- python3 bench.py --workload ar --steps 3 --warmup 1 --no-cpu-baseline --min-gpu-seconds 0 > gpurun_out/prof_ar.log 2>&1
echo "== PMC matrix-pipe utilisation (SQ counters + GRBM_GUI_ACTIVE, own passes)"
ISB_BENCH_INFLIGHT=1 ISB_HPE_LANES=1 timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_mfma_hpe -o run -- python3 bench.py --workload hpe --steps 1 --warmup 1 --no-cpu-baseline --no-extras --min-gpu-seconds 0 > gpurun_out/pmc_mfma_hpe.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_mfma_ar -o run -- python3 bench.py --workload ar --steps 1 --warmup 1 --no-cpu-baseline --no-extras --min-gpu-seconds 0 > gpurun_out/pmc_mfma_ar.log 2>&1
python3 tools/collect_mfma.py gpurun_out/mfma_hpe.json gpurun_out/pmc_mfma_hpe/run_counter_collection.csv
python3 tools/collect_mfma.py gpurun_out/mfma_ar.json gpurun_out/pmc_mfma_ar/run_counter_collection.csv
echo "== SQ counters of the three MFMA-bound early-stage kernels (VERDICT r3 item 5), own pass"
ISB_BENCH_INFLIGHT=1 ISB_HPE_LANES=1 timeout -k 10 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmc_lds_hpe -o run -- python3 bench.py --workload hpe --steps 1 --warmup 1 --no-cpu-baseline --no-extras --min-gpu-seconds 0 > gpurun_out/pmc_lds_hpe.log 2>&1 || true
python3 tools/collect_lds.py gpurun_out/lds_hpe.json gpurun_out/pmc_lds_hpe/run_counter_collection.csv || true
echo "== round 6: L2-side counters of a one-lane pose pass (VERDICT r5 item 4), the fused fronts alone (three forms, stamps, SQ / LDS counters)"
bash tools/pmc_l2.sh > gpurun_out/pmc_l2_run.log 2>&1 || true
PYTHONPATH=. timeout -k 10 300 python3 tools/exp_mbf16r.py 256 200 3 --stamps > gpurun_out/fronts_three_forms.txt 2>&1 || true
PYTHONPATH=. bash tools/pmc_any.sh fronts mbfront tools/exp_mbf16r.py 256 5 1 > gpurun_out/pmc_fronts_run.log 2>&1 || true
echo "== per-kernel A/B of the fused fronts' forms in the network (one-lane traces)"
bash tools/ab_kernels.sh ISB_MBF16_FORM 1 2 hpe > gpurun_out/ab_mbf16_forms.txt 2>&1 || true
bash tools/ab_kernels.sh ISB_MBF8_FORM 1 2 hpe > gpurun_out/ab_mbf8_forms.txt 2>&1 || true
echo done
