#!/bin/bash
# round 4 profiles, part B
R=r04
bash tools/profile_gpu.sh ${R}_update --workload j2_update > /dev/null || exit 1
bash tools/profile_gpu.sh ${R}_update_tangent --workload j2_update_tangent > /dev/null || exit 1
bash tools/profile_gpu.sh ${R}_hill_update_vjp --workload j2_update_vjp --yield-surface hill > /dev/null || exit 1
bash tools/profile_gpu.sh ${R}_uniaxial_update --workload j2_update --def-type uniaxial_stress --points 2000000 > /dev/null || exit 1
bash tools/profile_gpu.sh ${R}_hosford_update_tangent --workload hosford_update_tangent > /dev/null || exit 1
bash tools/profile_gpu.sh ${R}_ps_objective_grad --workload j2_objective_grad --def-type plane_stress > /dev/null || exit 1
ls gpurun_out/prof_${R}_*/summary_*.json
