#!/bin/bash
# full GPU suite + headline / pool benches on the library with the multi-layer networks
set -o pipefail
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r3_gpu_tests_deep.log 2>&1
echo "pytest exit $?" >> gpurun_out/r3_gpu_tests_deep.log
tail -3 gpurun_out/r3_gpu_tests_deep.log
for w in "" "--workload hosford_update" "--workload hybrid_update --points 5000000" "--workload j2_update_vjp --def-type plane_stress"; do
  timeout -k 10 120 python bench.py $w --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$w', '|', d['value'], d['unit'], '| ms', d['ms_per_step'], '| frac', d['roofline']['frac'])
" >> gpurun_out/r3_deep_bench.txt
done
cat gpurun_out/r3_deep_bench.txt
