#!/bin/bash
# kernel-level profile of the single-frame (stream) workload
set -e
R=$PWD
mkdir -p $R/gpurun_out/prof_stream
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stream -o stream -- python3 $R/bench.py --workload stream --steps 100 --warmup 10 --no-cpu-baseline > $R/gpurun_out/prof_stream/bench.log 2>&1
cd $R
f=$(ls gpurun_out/prof_stream/*kernel_stats.csv | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:25]:
    print(f"{r['Name'][:90]:90s} calls={r['Calls']:>7s} avg_us={float(r['AverageNs'])/1e3:8.2f} pct={r['Percentage']}")
PY
