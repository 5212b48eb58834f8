#!/bin/bash
# why are 3 lanes (86-frame lanes) 30 % slower than 2 since round 3? one switch at a time
run() { env "$@" ISB_HPE_LANES=3 timeout -k 10 200 python bench.py --workload hpe --steps 8 --warmup 3 --no-cpu-baseline --min-gpu-seconds 0 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; }
echo "default        $(run X=1)"
echo "F16=0          $(run ISB_HPE_F16=0)"
echo "DW_MAP8=0      $(run ISB_DW_MAP8=0)"
echo "batch 258 (2 lanes of 129) $(ISB_HPE_LANES=2 timeout -k 10 200 python bench.py --workload hpe --batch 258 --steps 8 --warmup 3 --no-cpu-baseline --min-gpu-seconds 0 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"
echo "batch 172 (2 lanes of 86)  $(ISB_HPE_LANES=2 timeout -k 10 200 python bench.py --workload hpe --batch 172 --steps 8 --warmup 3 --no-cpu-baseline --min-gpu-seconds 0 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"
echo "batch 192 (3 lanes of 64)  $(ISB_HPE_LANES=3 timeout -k 10 200 python bench.py --workload hpe --batch 192 --steps 8 --warmup 3 --no-cpu-baseline --min-gpu-seconds 0 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"
