#!/bin/bash
# Per-kernel A/B of one environment switch in one box session (one lane: every launch is a whole-batch launch):
#   bash tools/ab_kernels.sh VAR valueA valueB [workload]
# -> gpurun_out/abk_A/run_kernel_stats.csv, gpurun_out/abk_B/... and the per-kernel table (tools/kstat_diff.py)
var=$1; a=$2; b=$3; w=${4:-hpe}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for tag in A B; do
  if [ $tag = A ]; then export $var=$a; else export $var=$b; fi
  ISB_BENCH_INFLIGHT=1 ISB_HPE_LANES=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abk_$tag -o run -- python3 bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline --no-extras --min-gpu-seconds 0 > gpurun_out/abk_$tag.log 2>&1 || { tail -5 gpurun_out/abk_$tag.log; exit 1; }
done
python3 tools/kstat_diff.py gpurun_out/abk_A/run_kernel_stats.csv gpurun_out/abk_B/run_kernel_stats.csv
