#!/bin/bash
# final round-3 run: GPU suite + smoke, the rocprofv3 / PMC profiles of every bench workload, the side workloads, the default bench line
set -o pipefail
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r03_gpu_tests_final.log 2>&1
rc=$?; echo "pytest exit $rc" >> gpurun_out/r03_gpu_tests_final.log; tail -3 gpurun_out/r03_gpu_tests_final.log
[ $rc -eq 0 ] || exit 1
python -c "import __graft_entry__ as g; g.smoke()" || exit 1
bash tools/profile_sides.sh r03 || exit 1
bash tools/bench_sides.sh > gpurun_out/r03_side_workloads.txt 2>&1
python tools/objective_latency.py > gpurun_out/r03_objective_latency_final.txt 2>/dev/null
timeout -k 10 200 python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err && tail -c 400 gpurun_out/r03_bench_default.json
