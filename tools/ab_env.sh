#!/bin/bash
# A/B one environment switch in one session, interleaved: tools/ab_env.sh VAR valueA valueB [workload]
var=$1; a=$2; b=$3; w=${4:-hpe}
for i in 1 2 3; do for v in $a $b; do
  env $var=$v timeout -k 10 200 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-extras --min-gpu-seconds 0 > gpurun_out/abenv_$v$i.log 2>&1 || { tail -3 gpurun_out/abenv_$v$i.log; exit 1; }
  echo "$var=$v $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/abenv_$v$i.log) $(grep -o '"achieved": [0-9.]*' gpurun_out/abenv_$v$i.log)"
done; done
