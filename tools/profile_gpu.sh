#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: kernel-trace stats + separate PMC passes for bench.py.
# Usage: bash tools/profile_gpu.sh <tag> [extra bench args]
set -o pipefail
TAG=${1:-r01}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 250 timed launches back to back: the kernel-trace average then has a first-14 and a sustained (launch 50 on) part, parse_profiles.py
ARGS="--steps 250 --warmup 2 --no-cpu-baseline $@"
PARGS="--steps 3 --warmup 1 --no-cpu-baseline $@"
mix32() {   # the float side of the instruction mix (warm-start seeds run in float); not fatal if a counter is missing on this stack
  rocprofv3 --pmc SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU --output-format csv -d $OUT/pmc_mix32 -- python3 $ROOT/bench.py $PARGS > $OUT/bench_mix32.json 2> $OUT/mix32.err || echo "mix32 pass failed (see $OUT/mix32.err)"
}
if [ -n "$ONLY_MIX32" ]; then mix32; exit 0; fi     # (added to an earlier run of the other passes: re-run parse_profiles.py where all CSVs are)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py $PARGS > $OUT/bench_fetch.json 2> $OUT/fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py $PARGS > $OUT/bench_write.json 2> $OUT/write.err || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_WAVES --output-format csv -d $OUT/pmc_sq -- python3 $ROOT/bench.py $PARGS > $OUT/bench_sq.json 2> $OUT/sq.err || exit 1
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mix -- python3 $ROOT/bench.py $PARGS > $OUT/bench_mix.json 2> $OUT/mix.err || exit 1
mix32
cd $ROOT && python3 tools/parse_profiles.py $OUT $TAG
