#!/bin/bash
# round 4: uniaxial warm start -- tests + benches
O=gpurun_out/r04i; mkdir -p $O
SECONDS=0
timeout -k 10 1100 python -m pytest tests -q -m gpu > $O/tests_all.txt 2>&1; echo "all gpu tests rc=$? in ${SECONDS}s"; tail -6 $O/tests_all.txt
b() { python bench.py --no-cpu-baseline "$@" 2>>$O/bench.err | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); s=r.get('sustained',{}); print('$*', '| %.4g /s' % r['value'], '| kernel_ms %.4f' % r['roofline']['kernel_ms'], '| sustained ms', s.get('launches_50_250_ms'), 'frac', s.get('sustained_frac'))"; }
b --workload j2_update --def-type uniaxial_stress --points 2000000 --sustain
b --workload j2_update --def-type uniaxial_stress --points 2000000 --sustain --reference-iterates
b --workload j2_objective_grad --def-type uniaxial_stress --points 2000000 --sustain
b --workload j2_objective_grad --def-type uniaxial_stress --points 2000000 --sustain --reference-iterates
b --workload j2_update_vjp --def-type uniaxial_stress --yield-surface hill --points 2000000 --sustain
b --workload j2_update_vjp --def-type uniaxial_stress --yield-surface hill --points 2000000 --sustain --reference-iterates
