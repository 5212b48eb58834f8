#!/bin/bash
# Package power and shader clock while a bench workload runs (rocm-smi samples every 0.2 s beside a long run):
#   bash tools/power_probe.sh [workload=hpe] [steps=600]     -> gpurun_out/power_<workload>.txt (samples above 400 W = the timed region)
w=${1:-hpe}; n=${2:-600}
mkdir -p gpurun_out
( for i in $(seq 1 150); do rocm-smi --showpower --showclocks 2>/dev/null | grep -i "Package Power\|sclk" | sed 's/.*: //' | tr '\n' ' '; echo; sleep 0.2; done ) > gpurun_out/power_raw_$w.txt 2>&1 &
SM=$!
python bench.py --workload $w --steps $n --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', d['value'], d['unit'], d['ms_per_step'], 'ms per step over', d['steps'], 'steps')" | tee gpurun_out/power_$w.txt
kill $SM 2>/dev/null; wait $SM 2>/dev/null
python3 - "$w" <<'P' | tee -a gpurun_out/power_$w.txt
import re, sys
w = sys.argv[1]
pw, ck = [], []
for line in open(f"gpurun_out/power_raw_{w}.txt"):
    m = re.search(r"\((\d+)Mhz\)\s+([\d.]+)", line)
    if m and float(m.group(2)) > 400:
        ck.append(int(m.group(1))); pw.append(float(m.group(2)))
if pw:
    pw.sort(); ck.sort()
    print(f"{len(pw)} samples under load: package power min / median / max {pw[0]:.0f} / {pw[len(pw)//2]:.0f} / {pw[-1]:.0f} W; sclk as rocm-smi reports it {ck[0]} / {ck[len(ck)//2]} / {ck[-1]} MHz")
else:
    print("no sample under load")
P
rocm-smi --showmaxpower 2>/dev/null | grep -i "power (W)" | sed 's/.*: //' | xargs -I{} echo "package power cap {} W" | tee -a gpurun_out/power_$w.txt
