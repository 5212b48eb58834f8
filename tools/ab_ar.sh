#!/bin/bash
# A/B two builds in one session on the AR workload
for i in 1 2 3; do for v in A B; do
  ISB_LIB_PATH=$PWD/tools/ab/lib$v.so timeout -k 10 200 python bench.py --workload ar --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/abar_$v$i.log 2>&1 || { tail -3 gpurun_out/abar_$v$i.log; exit 1; }
  echo "$v$i $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/abar_$v$i.log) $(grep -o '"achieved": [0-9.]*' gpurun_out/abar_$v$i.log)"
done; done
