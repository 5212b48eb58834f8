#!/bin/bash
# A/B: the LDS-staged 8x8 depthwise kernel (ISB_DW_MAP8) and the ROI-only host input (ISB_HPE_ROI), same session
for m in 1 0; do for i in 1 2; do
  ISB_HPE_ROI=$m timeout -k 10 200 python bench.py --workload hpe --host-input --steps 10 --warmup 3 --no-cpu-baseline --min-gpu-seconds 0 > gpurun_out/roi_$m$i.log 2>&1 || { tail -3 gpurun_out/roi_$m$i.log; exit 1; }
  echo "ISB_HPE_ROI=$m host-input $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/roi_$m$i.log) $(grep -o '"value": [0-9.]*' gpurun_out/roi_$m$i.log | head -1)"
done; done
