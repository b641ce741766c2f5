#!/bin/bash
# extra PMC passes on the Hosford pool kernel: where do its wait cycles come from?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
A="--workload hosford_update --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d $R/gpurun_out/pmc_hos1 -- python3 $R/bench.py $A > /dev/null 2> $R/gpurun_out/pmc_hos1.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_IFETCH SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SMEM --output-format csv -d $R/gpurun_out/pmc_hos2 -- python3 $R/bench.py $A > /dev/null 2> $R/gpurun_out/pmc_hos2.err
cd $R
python3 - <<'PY'
import csv, glob, collections
for d in ('gpurun_out/pmc_hos1','gpurun_out/pmc_hos2'):
    f=glob.glob(d+'/*/*_counter_collection.csv')[0]
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'k_update_pool' in r['Kernel_Name'] and int(r['Grid_Size'])>100000:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items(): print(d.split('/')[-1], k, sum(v)/len(v), len(v))
PY
