#!/bin/bash
# SQ / LDS counters of the isb:: kernels ANY python tool launches, one rocprofv3 --pmc pass per counter set (no tracing domains beside it).
# usage: tools/pmc_any.sh <tag> <kernel substring> tools/<script>.py [args...]      -> gpurun_out/pmc_<tag>/summary.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
match=$1; shift
mkdir -p gpurun_out/pmc_$tag
i=0
# PMC_SETS="set one;set two;..." replaces the default SQ / LDS sets (e.g. the L2-side sets of tools/pmc_l2.sh)
if [ -n "$PMC_SETS" ]; then
  IFS=';' read -ra SETS <<< "$PMC_SETS"
  for set in "${SETS[@]}"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_$tag -o set$i -- python3 "$@" > gpurun_out/pmc_$tag/set$i.log 2>&1 || { tail -3 gpurun_out/pmc_$tag/set$i.log; }
  done
else
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SMEM SQ_VALU_MFMA_COEXEC_CYCLES" \
           "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_MEM_VIOLATIONS SQ_INSTS_LDS_DMA"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_$tag -o set$i -- python3 "$@" > gpurun_out/pmc_$tag/set$i.log 2>&1 || { tail -3 gpurun_out/pmc_$tag/set$i.log; }
done
fi
python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in glob.glob('gpurun_out/pmc_$tag/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if '$match' not in k: continue
        k=k[:90]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[(k,r['Counter_Name'])]+=1
with open('gpurun_out/pmc_$tag/summary.txt','w') as out:
    for k,d in agg.items():
        print(k, file=out); print(k)
        for c,v in sorted(d.items()):
            line='   %-28s %16.0f (per dispatch, %d dispatches)'%(c, v/max(1,cnt[(k,c)]), cnt[(k,c)])
            print(line, file=out); print(line)
PY
