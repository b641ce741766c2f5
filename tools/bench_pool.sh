#!/bin/bash
# work-pool kernel (default) against CM_SOLVER_LOCKSTEP on the iteration-bound update workloads
run() { python bench.py --no-cpu-baseline --steps 5 --warmup 2 "$@" 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('$*', '| %.4g updates/s' % r['value'], '| ms_per_step %.4f' % r['ms_per_step'], '| frac %.3f' % r['roofline']['frac'])"; }
run --workload hosford_update
run --workload hosford_update --lockstep
run --workload hybrid_update --points 5000000
run --workload hybrid_update --points 5000000 --lockstep
run --workload j2_update --def-type plane_stress
run --workload j2_update --def-type plane_stress --lockstep
run --workload j2_update --yield-surface hill
run --workload j2_update --yield-surface hill --lockstep
run --workload j2_update --yield-surface hosford8
run --workload j2_update --yield-surface hosford8 --lockstep
run --workload j2_update --yield-surface barlat8 --points 2000000
run --workload j2_update --yield-surface barlat8 --points 2000000 --lockstep
