#!/bin/bash
# rocprofv3 kernel-trace stats + PMC passes (HBM traffic, SQ busy / wait, fp64 instruction mix) for the headline and the side
# workloads of DESIGN.md section 6.  Run ON THE GPU BOX from the repo root: bash tools/profile_sides.sh <round-tag>
R=${1:-r03}
bash tools/profile_gpu.sh ${R}_headline > /dev/null || exit 1
bash tools/profile_gpu.sh ${R}_hosford --workload hosford_update > /dev/null || exit 1
bash tools/profile_gpu.sh ${R}_hybrid --workload hybrid_update --points 5000000 > /dev/null || exit 1
bash tools/profile_gpu.sh ${R}_ps_update_vjp --workload j2_update_vjp --def-type plane_stress > /dev/null || exit 1
bash tools/profile_gpu.sh ${R}_objective_grad --workload j2_objective_grad > /dev/null || exit 1
bash tools/profile_gpu.sh ${R}_barlat --workload j2_update --yield-surface barlat8 --points 2000000 > /dev/null || exit 1
bash tools/profile_gpu.sh ${R}_update --workload j2_update > /dev/null || exit 1
bash tools/profile_gpu.sh ${R}_update_tangent --workload j2_update_tangent > /dev/null || exit 1
bash tools/profile_gpu.sh ${R}_hill_update_vjp --workload j2_update_vjp --yield-surface hill > /dev/null || exit 1
bash tools/profile_gpu.sh ${R}_uniaxial_update --workload j2_update --def-type uniaxial_stress --points 2000000 > /dev/null || exit 1
bash tools/profile_gpu.sh ${R}_hosford_update_tangent --workload hosford_update_tangent > /dev/null || exit 1
bash tools/profile_gpu.sh ${R}_ps_objective_grad --workload j2_objective_grad --def-type plane_stress > /dev/null || exit 1
ls gpurun_out/prof_${R}_*/summary_*.json
