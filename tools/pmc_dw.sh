#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
for set in "FETCH_SIZE WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVES" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 120 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_$tag -o $n -- python3 tools/dw_one.py "$@" > gpurun_out/pmc_$tag.log 2>&1 || { tail -3 gpurun_out/pmc_$tag.log; }
done
python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in glob.glob('gpurun_out/pmc_$tag/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'dwconv' not in r['Kernel_Name']: continue
        agg[r['Kernel_Name'][:60]][r['Counter_Name']]+=float(r['Counter_Value']); cnt[(r['Kernel_Name'][:60],r['Counter_Name'])]+=1
for k,d in agg.items():
    print(k)
    for c,v in sorted(d.items()): print('   %-28s %16.0f (per dispatch)'%(c, v/max(1,cnt[(k,c)])))
PY
