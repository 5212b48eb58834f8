#!/bin/bash
# stream (B = 1) latency with and without the one-launch Fused-MBConv blocks
for r in 1 2; do for f in 0 1; do
  ISB_FUSE_BLOCK=$f timeout -k 10 200 python bench.py --workload stream --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/fbs_$f.log 2>&1 || { tail -5 gpurun_out/fbs_$f.log; exit 1; }
  echo "stream fuse_block $f: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/fbs_$f.log)"
done; done
