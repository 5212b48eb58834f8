#!/bin/bash
# A/B one bench.py argument in one session, interleaved: tools/ab_arg.sh --flag valueA valueB [workload] [extra args...]
flag=$1; a=$2; b=$3; w=${4:-hpe}; shift 4
for i in 1 2 3; do for v in $a $b; do
  timeout -k 10 200 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-extras --min-gpu-seconds 0 $flag $v "$@" > gpurun_out/abarg_$v$i.log 2>&1 || { tail -3 gpurun_out/abarg_$v$i.log; exit 1; }
  echo "$flag=$v $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/abarg_$v$i.log) $(grep -o '"achieved": [0-9.]*' gpurun_out/abarg_$v$i.log)"
done; done
