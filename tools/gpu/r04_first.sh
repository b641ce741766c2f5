#!/bin/bash
# round 4, first lease: the bench launcher through torch.distributed.run + sustained figures of the round-3 kernels
set -o pipefail
mkdir -p gpurun_out/r04a
python -m pytest tests/test_bench_launcher.py -m gpu -x -q > gpurun_out/r04a/launcher_test.txt 2>&1; echo "launcher test rc=$?" 
tail -3 gpurun_out/r04a/launcher_test.txt
CMAD_BENCH_FORCE_DIST=1 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04a/bench_dist1.json 2> gpurun_out/r04a/bench_dist1.err; echo "bench dist rc=$?"
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04a/bench.json 2> gpurun_out/r04a/bench.err; echo "bench rc=$?"
tools/bench_sustain.sh > gpurun_out/r04a/sustained_r03_kernels.txt 2>&1
cat gpurun_out/r04a/sustained_r03_kernels.txt
