#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_ar -o $n -- python3 bench.py --workload ar --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_ar.log 2>&1 || { tail -3 gpurun_out/pmc_ar.log; }
done
python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in glob.glob('gpurun_out/pmc_ar/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:50]
        if 'ar_proto' not in k and 'ar_stats' not in k: continue
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[(k,r['Counter_Name'])]+=1
for k,d in agg.items():
    print(k)
    for c,v in sorted(d.items()): print('   %-28s %16.0f (per dispatch, %d dispatches)'%(c, v/max(1,cnt[(k,c)]), cnt[(k,c)]))
PY
