#!/bin/bash
# do three / four lanes lose to stream -> hardware-queue multiplexing? (GPU_MAX_HW_QUEUES, default 4)
run() { env "$@" timeout -k 10 200 python bench.py --workload hpe --steps 8 --warmup 3 --no-cpu-baseline --no-extras --min-gpu-seconds 0 --batch $B 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; }
for B in 128 256; do for q in 4 8 16; do for l in 2 3 4; do echo "batch=$B queues=$q lanes=$l $(run GPU_MAX_HW_QUEUES=$q ISB_HPE_LANES=$l)"; done; done; done
