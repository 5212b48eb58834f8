#!/bin/bash
for r in 1 2 3; do for f in 256 384; do
  ISB_FUSE_BLOCK_CEXP=$f timeout -k 10 200 python bench.py --workload hpe --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/fb_$f.log 2>&1 || { tail -5 gpurun_out/fb_$f.log; exit 1; }
  echo "fuse up to cexp $f: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/fb_$f.log) $(grep -o '"achieved": [0-9.]*' gpurun_out/fb_$f.log)"
done; done
