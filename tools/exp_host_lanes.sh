#!/bin/bash
# host-input rate against the number of lanes: every lane gathers its own frames, so only the FIRST lane's transfer is exposed
for l in 2 3 4; do for hi in 1 0; do
  extra=""; [ $hi = 1 ] && extra="--host-input"
  ISB_HPE_LANES=$l timeout -k 10 200 python bench.py --workload hpe $extra --steps 10 --warmup 3 --no-cpu-baseline --min-gpu-seconds 0 > gpurun_out/hl_$l$hi.log 2>&1 || { tail -3 gpurun_out/hl_$l$hi.log; exit 1; }
  echo "lanes=$l host_input=$hi $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/hl_$l$hi.log) $(grep -o '"value": [0-9.]*' gpurun_out/hl_$l$hi.log | head -1)"
done; done
