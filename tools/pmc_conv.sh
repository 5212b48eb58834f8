#!/bin/bash
# usage: tools/pmc_conv.sh <tag> <hw> <cin> <cout> <k> <variant>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 120 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_$tag -o $n -- python3 tools/conv_one.py "$@" > gpurun_out/pmc_$tag.log 2>&1 || { tail -3 gpurun_out/pmc_$tag.log; }
done
python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in glob.glob('gpurun_out/pmc_$tag/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'conv_igemm' not in r['Kernel_Name']: continue
        agg[r['Kernel_Name'][:60]][r['Counter_Name']]+=float(r['Counter_Value']); cnt[(r['Kernel_Name'][:60],r['Counter_Name'])]+=1
for k,d in agg.items():
    print(k)
    for c,v in sorted(d.items()): print('   %-28s %14.0f (per dispatch)'%(c, v/max(1,cnt[(k,c)])))
PY
