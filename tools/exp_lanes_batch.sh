#!/bin/bash
# frames per lane vs lanes: does running smaller lanes side by side beat one larger lane? (ms per step)
run() { env "$@" timeout -k 10 200 python bench.py --workload hpe --steps 8 --warmup 3 --no-cpu-baseline --no-extras --min-gpu-seconds 0 --batch $B 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; }
for B in 64 128 256; do for l in 1 2 4; do echo "batch=$B lanes=$l $(run ISB_HPE_LANES=$l)"; done; done
