#!/bin/bash
# same-session comparison of several values of one environment switch, interleaved: tools/ab_envn.sh VAR "v1 v2 v3 ..." [workload] [rounds]
var=$1; vals=$2; w=${3:-hpe}; n=${4:-3}
for i in $(seq 1 $n); do for v in $vals; do
  env $var=$v timeout -k 10 200 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-extras --min-gpu-seconds 0 > gpurun_out/abenvn_$v$i.log 2>&1 || { tail -3 gpurun_out/abenvn_$v$i.log; exit 1; }
  echo "$var=$v $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/abenvn_$v$i.log) $(grep -o '"achieved": [0-9.]*' gpurun_out/abenvn_$v$i.log | head -1)"
done; done
