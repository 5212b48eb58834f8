#!/bin/bash
# match stage of step k beside the pose stage of step k + 1 (ISB_BENCH_OVERLAP=2) x hardware queues
run() { env "$@" timeout -k 10 200 python bench.py --workload pipeline --steps 10 --warmup 3 --no-cpu-baseline --no-extras --min-gpu-seconds 0 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; }
for i in 1 2; do for q in 4 8; do for o in 1 2; do echo "queues=$q overlap=$o $(run GPU_MAX_HW_QUEUES=$q ISB_BENCH_OVERLAP=$o)"; done; done; done
