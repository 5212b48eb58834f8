#!/bin/bash
# usage: tools/run_grid_exp.sh  (on the GPU box) -- HPE bench under each ISB_CONV_GRID mode
mkdir -p gpurun_out
for m in 0 1 2; do
  ISB_CONV_GRID=$m timeout -k 10 200 python bench.py --workload hpe --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/grid_$m.log 2>&1 || exit 1
  echo "mode $m: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/grid_$m.log) $(grep -o '"achieved": [0-9.]*' gpurun_out/grid_$m.log)"
done
