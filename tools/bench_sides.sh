#!/bin/bash
# side measurements for DESIGN.md section 6: every bench workload once, 10 steps
run() { python bench.py --no-cpu-baseline --steps 10 "$@" 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('$*', '| %.4g updates/s' % r['value'], '| kernel_ms %.4f' % r['roofline']['kernel_ms'], '| frac %.3f' % r['roofline']['frac'])"; }
run --workload j2_update_vjp
run --workload j2_update_vjp --ls-evals 4
run --workload j2_update
run --workload j2_update_tangent
run --workload j2_update_tangent --def-type plane_stress
run --workload j2_update --ls-evals 4
run --workload j2_objective_grad
run --workload j2_objective_grad --ls-evals 4
run --workload j2_update_vjp --def-type plane_stress
run --workload j2_update_vjp --def-type plane_stress --ls-evals 4
run --workload j2_update --def-type plane_stress
run --workload j2_objective_grad --def-type plane_stress
run --workload hosford_update --steps 5
run --workload hybrid_update --points 5000000 --steps 5
run --workload j2_update_vjp --general-newton
run --workload j2_update --general-newton
run --workload j2_objective_grad --general-newton
run --workload ps_calibration_history --steps 5
run --workload ps_calibration_history --steps 5 --per-step-history
run --workload j2_update_vjp --yield-surface hill
run --workload j2_update_vjp --yield-surface hill --def-type plane_stress
run --workload j2_update_vjp --yield-surface hosford8
run --workload j2_update_vjp --yield-surface barlat8 --points 2000000
run --workload hosford_update_vjp --steps 5
run --workload hosford_update_tangent --steps 5
run --workload hybrid_update_vjp --points 5000000 --steps 5
run --workload hybrid_update_tangent --points 5000000 --steps 5
run --workload calibration_history --steps 5
run --workload j2_update --def-type uniaxial_stress --points 2000000
run --workload j2_objective_grad --def-type uniaxial_stress --points 2000000
