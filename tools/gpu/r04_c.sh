#!/bin/bash
# round 4, lease 3: warm starts (Hill FULL_3D, J2 PLANE_STRESS, Hosford) + screened route -- parity tests, then side benches
set -o pipefail
O=gpurun_out/r04c; mkdir -p $O
python -m pytest tests/test_gpu_update.py -x -q -m gpu > $O/tests_update.txt 2>&1; echo "update tests rc=$?"; tail -4 $O/tests_update.txt
python -m pytest tests/test_gpu_pool.py -x -q -m gpu > $O/tests_pool.txt 2>&1; echo "pool tests rc=$?"; tail -4 $O/tests_pool.txt
b() { python bench.py --no-cpu-baseline "$@" 2>>$O/bench.err | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); s=r.get('sustained',{}); print('$*', '| %.4g /s' % r['value'], '| kernel_ms %.4f' % r['roofline']['kernel_ms'], '| frac %.3f' % r['roofline']['frac'], '| sustained ms', s.get('launches_50_250_ms'), 'frac', s.get('sustained_frac'))"; }
b --workload hosford_update --sustain
b --workload hosford_update_vjp --sustain
b --workload j2_update_vjp --yield-surface hill --sustain
b --workload j2_update_vjp --yield-surface hill --sustain --reference-iterates
b --workload j2_update --yield-surface hill --sustain
b --workload j2_objective_grad --yield-surface hill --sustain
b --workload j2_update_vjp --def-type plane_stress --sustain
b --workload j2_update_vjp --def-type plane_stress --sustain --reference-iterates
b --workload j2_objective_grad --def-type plane_stress --sustain
b --workload j2_objective_grad --def-type plane_stress --sustain --reference-iterates
b --workload j2_update --def-type plane_stress --sustain
b --workload hybrid_update --points 5000000 --sustain
CM_DEBUG_NO_SCREEN=1 python bench.py --no-cpu-baseline --workload hybrid_update --points 5000000 --sustain 2>>$O/bench.err | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); s=r['sustained']; print('hybrid_update NO_SCREEN (pool) sustained ms', s['launches_50_250_ms'])"
b --workload hybrid_update_vjp --points 5000000 --sustain
b --workload hybrid_update_tangent --points 5000000 --steps 10
b --workload j2_update --yield-surface barlat8 --points 2000000 --sustain
CM_DEBUG_NO_SCREEN=1 python bench.py --no-cpu-baseline --workload j2_update --yield-surface barlat8 --points 2000000 --sustain 2>>$O/bench.err | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); s=r['sustained']; print('barlat8 update NO_SCREEN (lockstep) sustained ms', s['launches_50_250_ms'])"
b --workload hosford_update --general-newton --sustain
CM_DEBUG_NO_SCREEN=1 python bench.py --no-cpu-baseline --workload hosford_update --general-newton --sustain 2>>$O/bench.err | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); s=r['sustained']; print('hosford reference iteration NO_SCREEN (pool) sustained ms', s['launches_50_250_ms'])"
