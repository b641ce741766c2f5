#!/bin/bash
# full GPU suite + the bench workloads after the library shrink (one Newton loop for plain and line-search use in the cold configurations)
set -o pipefail
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r3_gpu_tests_shrink.log 2>&1
echo "pytest exit $?" >> gpurun_out/r3_gpu_tests_shrink.log
tail -3 gpurun_out/r3_gpu_tests_shrink.log
rm -f gpurun_out/r3_shrink_bench.txt
for w in "" "--workload j2_update" "--workload j2_update_tangent" "--workload j2_objective_grad" "--workload hosford_update" "--workload hosford_update_vjp" "--workload hybrid_update --points 5000000" "--workload j2_update_vjp --def-type plane_stress" "--workload j2_update_vjp --yield-surface hill" "--workload j2_update_vjp --yield-surface hosford8" "--workload j2_update --yield-surface barlat8 --points 2000000" "--workload j2_update --def-type uniaxial_stress"; do
  timeout -k 10 120 python bench.py $w --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$w', '|', '%.4g' % d['value'], d['unit'], '| ms', '%.4f' % d['ms_per_step'], '| frac', '%.3f' % d['roofline']['frac'])
" >> gpurun_out/r3_shrink_bench.txt
done
cat gpurun_out/r3_shrink_bench.txt
