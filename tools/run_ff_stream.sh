#!/bin/bash
# stream (B = 1) latency with the fused expand + depthwise + pool front (ISB_FUSE_FRONT=1) vs the two-launch form
mkdir -p gpurun_out
for r in 1 2; do for f in 0 1; do
  if [ $f = 1 ]; then export ISB_FUSE_FRONT=1; else unset ISB_FUSE_FRONT; fi
  timeout -k 10 200 python bench.py --workload stream --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/ffs_$f.log 2>&1 || { tail -5 gpurun_out/ffs_$f.log; exit 1; }
  echo "stream fuse_front $f: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/ffs_$f.log)"
done; done
