#!/bin/bash
mkdir -p gpurun_out
for i in 1 2; do for l in 1 2 3; do
  ISB_HPE_LANES=$l timeout -k 10 200 python bench.py --workload hpe --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/lanes_$l.log 2>&1 || exit 1
  echo "lanes=$l $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/lanes_$l.log)"
done; done
