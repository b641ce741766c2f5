#!/bin/bash
# extra PMC passes for the bench kernel (instruction mix, busy cycles); run on the GPU box from the repo root
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/pmc_extra; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
A="--steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_SALU SQ_INSTS_VALU --output-format csv -d $OUT/a -- python3 $ROOT/bench.py $A > /dev/null 2> $OUT/a.err
rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_ACTIVE_INST_VALU SQ_LEVEL_WAVES SQ_WAVE_CYCLES SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM --output-format csv -d $OUT/b -- python3 $ROOT/bench.py $A > /dev/null 2> $OUT/b.err
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/c -- python3 $ROOT/bench.py $A > /dev/null 2> $OUT/c.err
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC --output-format csv -d $OUT/d -- python3 $ROOT/bench.py $A > /dev/null 2> $OUT/d.err
cd $ROOT && python3 - <<'PY'
import csv, glob, collections
for sub in "abcd":
    acc = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/pmc_extra/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_reverse" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(sub, {k: round(sum(v)/len(v)) for k, v in acc.items()})
PY
