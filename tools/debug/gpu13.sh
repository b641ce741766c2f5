#!/bin/bash
# where does ps_calibration_history spend its time?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_pshist -- python3 $R/bench.py --workload ps_calibration_history --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/pshist.json 2> $R/gpurun_out/pshist.err
cd $R
f=$(ls gpurun_out/prof_pshist/*/*_kernel_stats.csv | head -1)
cut -c1-200 $f | head -8
tail -1 gpurun_out/pshist.json | cut -c1-300
python bench.py --workload calibration_history --def-type plane_stress --steps 5 --no-cpu-baseline 2>/dev/null | cut -c1-400
python bench.py --workload ps_calibration_history --steps 5 --no-cpu-baseline --general-newton 2>/dev/null | cut -c1-400
