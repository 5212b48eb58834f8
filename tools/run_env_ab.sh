#!/bin/bash
# A/B one environment switch on the HPE workload, interleaved: tools/run_env_ab.sh VAR [workload]
V=$1; W=${2:-hpe}
mkdir -p gpurun_out
for i in 1 2 3; do for f in 0 1; do
  env $V=$f timeout -k 10 200 python bench.py --workload $W --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/env_$f.log 2>&1 || { tail -3 gpurun_out/env_$f.log; exit 1; }
  echo "$V=$f $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/env_$f.log)"
done; done
