#!/bin/bash
# what do the SE-gated projections pull through the L2 -> CU ports? TCP / TCC counters of every isb:: kernel of a one-lane
# 256-frame pose pass (own --pmc passes, no tracing domain beside them), then bytes per clock and CU for the kernels named on the command line.
#   bash tools/pmc_l2.sh          -> gpurun_out/pmc_l2/summary.txt + gpurun_out/pmc_l2/l2_table.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export ISB_BENCH_INFLIGHT=1 ISB_HPE_LANES=1
PMC_SETS="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum;TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum;TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum;TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum;GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVES" \
  bash tools/pmc_any.sh l2 "isb::" bench.py --workload hpe --steps 2 --warmup 1 --no-cpu-baseline --no-extras --min-gpu-seconds 0 > gpurun_out/pmc_l2.log 2>&1
python3 - <<'PY'
import re
txt = open('gpurun_out/pmc_l2/summary.txt').read()
rows = []
for b in re.split(r'\n(?=void |[a-z_]+_kernel)', txt):
    lines = b.strip().split('\n')
    d = {}
    for l in lines[1:]:
        m = re.match(r'\s+(\S+)\s+(\d+)', l)
        if m:
            d[m.group(1)] = float(m.group(2))
    if 'GRBM_GUI_ACTIVE' not in d or 'TCP_TCC_READ_REQ_sum' not in d:
        continue
    cyc = d['GRBM_GUI_ACTIVE'] / 8.0                       # the counter sums the 8 XCDs
    rd = d['TCP_TCC_READ_REQ_sum']
    rows.append((cyc, lines[0][:78], rd, d.get('TCC_HIT_sum', 0), d.get('TCC_MISS_sum', 0), d.get('TCC_EA0_RDREQ_sum', 0), d.get('TCP_TCC_READ_REQ_LATENCY_sum', 0),
                 d.get('TCP_PENDING_STALL_CYCLES_sum', 0), d.get('TCP_TOTAL_ACCESSES_sum', 0)))
with open('gpurun_out/pmc_l2/l2_table.txt', 'w') as out:
    hdr = f"{'kernel':78s} {'cycles':>9s} {'L1->L2 read req':>16s} {'B/clk/CU @64B':>14s} {'@128B':>7s} {'L2 hit':>7s} {'EA rdreq':>10s} {'lat/req':>8s} {'TCP stall/cyc/CU':>17s}"
    print(hdr); print(hdr, file=out)
    for cyc, name, rd, hit, miss, ea, lat, stall, acc in sorted(rows, reverse=True)[:40]:
        line = f"{name:78s} {cyc:9.0f} {rd:16.0f} {rd * 64 / cyc / 256:14.1f} {rd * 128 / cyc / 256:7.1f} {hit / max(hit + miss, 1):7.3f} {ea:10.0f} {lat / max(rd, 1):8.0f} {stall / cyc / 256:17.2f}"
        print(line); print(line, file=out)
PY
