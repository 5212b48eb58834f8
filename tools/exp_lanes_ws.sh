#!/bin/bash
# lanes x the persistent weights-stationary expand kernels (ISB_WSREG=0: tile kernels) x the loader-wave projections
run() { env "$@" timeout -k 10 200 python bench.py --workload hpe --steps 8 --warmup 3 --no-cpu-baseline --no-extras --min-gpu-seconds 0 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; }
for l in 2 3 4; do
  echo "lanes=$l default          $(run ISB_HPE_LANES=$l)"
  echo "lanes=$l WSREG=0          $(run ISB_HPE_LANES=$l ISB_WSREG=0)"
  echo "lanes=$l LW=0             $(run ISB_HPE_LANES=$l ISB_LW=0)"
  echo "lanes=$l WSREG=0 LW=0     $(run ISB_HPE_LANES=$l ISB_WSREG=0 ISB_LW=0)"
done
