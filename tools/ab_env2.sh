#!/bin/bash
# A/B one environment switch with a fixed second one: tools/ab_env2.sh VAR a b FIXEDVAR=value [workload]
var=$1; a=$2; b=$3; fixed=$4; w=${5:-hpe}
for i in 1 2 3; do for v in $a $b; do
  env $fixed $var=$v timeout -k 10 200 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-extras --min-gpu-seconds 0 > gpurun_out/abenv2_$v$i.log 2>&1 || { tail -3 gpurun_out/abenv2_$v$i.log; exit 1; }
  echo "$fixed $var=$v $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/abenv2_$v$i.log) $(grep -o '"achieved": [0-9.]*' gpurun_out/abenv2_$v$i.log)"
done; done
