#!/bin/bash
# dynamic half assignment of the pool kernel: parity tests on the variant library, then same-box A/B against static 32-point chunks
export CMAD_HIP_LIB=ab_libs/dyn.so
timeout -k 10 600 python -m pytest tests/test_gpu_update.py -m gpu -x -q -k "full_size or work_pool or hosford or hybrid or edge" > gpurun_out/r03_dyn_tests.log 2>&1
echo "pytest exit $?" >> gpurun_out/r03_dyn_tests.log; tail -3 gpurun_out/r03_dyn_tests.log
unset CMAD_HIP_LIB
ab() { for rep in 1 2; do for L in ab_libs/static32.so ab_libs/dyn.so; do
  CMAD_HIP_LIB=$L python bench.py --no-cpu-baseline $1 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('$L', '$1', '| %.4g' % r['value'], '| kernel_ms %.4f' % r['roofline']['kernel_ms'])"; done; done; }
ab "--workload hosford_update --steps 10 --warmup 3" > gpurun_out/r03_pool_dynamic_ab.txt
ab "--workload hybrid_update --points 5000000 --steps 10 --warmup 3" >> gpurun_out/r03_pool_dynamic_ab.txt
ab "--workload hosford_update_tangent --steps 5 --warmup 2" >> gpurun_out/r03_pool_dynamic_ab.txt
ab "--workload hybrid_update_vjp --points 5000000 --steps 5 --warmup 2" >> gpurun_out/r03_pool_dynamic_ab.txt
cat gpurun_out/r03_pool_dynamic_ab.txt
