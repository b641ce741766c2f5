#!/bin/bash
# sustained-clock figures: 250 back-to-back launches after an idle gap per workload; prints the mean of the
# first 14 launches, the mean of launches 50-250 and a decimated per-launch trace.  Usage: tools/bench_sustain.sh > profiles/rNN_sustained.txt
run() {
  python bench.py --no-cpu-baseline --sustain "$@" 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); s=r['sustained']; t=s['per_launch_ms']
print('$*')
print('   first 14 launches: %.4f ms = %.3f of the HBM roof | launches 50-250: %.4f ms = %.3f | sustained %.4g updates/s' % (s['first_14_ms'], s['first_14_frac'], s['launches_50_250_ms'], s['sustained_frac'], s['sustained_value']))
print('   per-launch ms (launch 0,1,2,3,5,8,13,20,30,50,100,150,200,249):', ' '.join('%.4f' % t[i] for i in (0,1,2,3,5,8,13,20,30,50,100,150,200,249)))
"; }
run --workload j2_update_vjp
run --workload j2_objective_grad
run --workload j2_update
run --workload j2_update_vjp --general-newton
run --workload j2_update_vjp --def-type plane_stress
run --workload j2_objective_grad --def-type plane_stress
run --workload j2_update_vjp --yield-surface hill
run --workload hosford_update
run --workload hybrid_update --points 5000000
run --workload j2_update_vjp --yield-surface barlat8 --points 2000000
run --workload j2_update --def-type uniaxial_stress --points 2000000
run --workload j2_update_vjp --def-type plane_stress --reference-iterates
run --workload j2_objective_grad --def-type plane_stress --reference-iterates
run --workload j2_update_vjp --yield-surface hill --reference-iterates
run --workload hosford_update --general-newton
run --workload hosford_update_vjp
run --workload hybrid_update_vjp --points 5000000
run --workload j2_update --def-type uniaxial_stress --points 2000000 --reference-iterates
run --workload j2_objective_grad --def-type uniaxial_stress --points 2000000
run --workload j2_update_vjp --def-type uniaxial_stress --yield-surface hill --points 2000000
run --workload j2_update_vjp --def-type plane_stress --yield-surface hill
run --workload j2_update_vjp --def-type plane_stress --yield-surface hill --reference-iterates
