#!/bin/bash
mkdir -p gpurun_out
for r in 1 2; do for l in 1 2 3 4; do
  ISB_HPE_LANES=$l timeout -k 10 200 python bench.py --workload hpe --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/lane_${l}.log 2>&1 || { tail -5 gpurun_out/lane_${l}.log; exit 1; }
  echo "lanes $l: $(grep -o '"value": [0-9.]*' gpurun_out/lane_${l}.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/lane_${l}.log) $(grep -o '"achieved": [0-9.]*' gpurun_out/lane_${l}.log)"
done; done
