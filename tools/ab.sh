#!/bin/bash
# A/B two builds in one session: tools/ab/libA.so vs tools/ab/libB.so, interleaved, HPE workload
for i in 1 2 3; do for v in A B; do
  ISB_LIB_PATH=$PWD/tools/ab/lib$v.so timeout -k 10 200 python bench.py --workload hpe --steps 10 --warmup 3 --no-cpu-baseline --no-extras --min-gpu-seconds 0 > gpurun_out/ab_$v$i.log 2>&1 || { tail -3 gpurun_out/ab_$v$i.log; exit 1; }
  echo "$v$i $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/ab_$v$i.log) $(grep -o '"achieved": [0-9.]*' gpurun_out/ab_$v$i.log)"
done; done
