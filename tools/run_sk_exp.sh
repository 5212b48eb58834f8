#!/bin/bash
# stream (B = 1) latency with and without split-K on the single-frame projections
mkdir -p gpurun_out
for r in 1 2; do for f in 0 1; do
  ISB_SPLIT_K=$f timeout -k 10 200 python bench.py --workload stream --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/sk_$f.log 2>&1 || { tail -5 gpurun_out/sk_$f.log; exit 1; }
  echo "stream split_k $f: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/sk_$f.log)"
done; done
