#!/bin/bash
# A/B two builds of the library in one session, interleaved: tools/ab_lib.sh <old.so> [workload] [extra bench args...]
old=$1; w=${2:-hpe}; shift 2
for i in 1 2 3; do for v in old new; do
  if [ $v = old ]; then export ISB_LIB_PATH=$old; else unset ISB_LIB_PATH; fi
  timeout -k 10 200 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-extras --min-gpu-seconds 0 "$@" > gpurun_out/ablib_$v$i.log 2>&1 || { tail -3 gpurun_out/ablib_$v$i.log; exit 1; }
  echo "$v $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/ablib_$v$i.log) $(grep -o '"achieved": [0-9.]*' gpurun_out/ablib_$v$i.log)"
done; done
