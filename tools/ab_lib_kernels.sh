#!/bin/bash
# Per-kernel A/B of two builds of the library in one box session (one lane: every launch is a whole-batch launch):
#   bash tools/ab_lib_kernels.sh <old.so> [workload]   -> gpurun_out/ablk_old, gpurun_out/ablk_new + the per-kernel table
old=$1; w=${2:-hpe}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for tag in old new; do
  if [ $tag = old ]; then export ISB_LIB_PATH=$old; else unset ISB_LIB_PATH; fi
  ISB_BENCH_INFLIGHT=1 ISB_HPE_LANES=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ablk_$tag -o run -- python3 bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline --no-extras --min-gpu-seconds 0 > gpurun_out/ablk_$tag.log 2>&1 || { tail -5 gpurun_out/ablk_$tag.log; exit 1; }
done
python3 tools/kstat_diff.py gpurun_out/ablk_old/run_kernel_stats.csv gpurun_out/ablk_new/run_kernel_stats.csv
