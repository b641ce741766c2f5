#!/bin/bash
# rocprofv3 kernel-trace stats + PMC passes (tools/profile_gpu.sh) of the twelve bench workloads of DESIGN.md section 6, in two
# leases (six workloads each fit one gpurun call):
#   gpurun --timeout 1190 -- 'bash tools/gpu/profile.sh <round tag> a'      then      ... b
# afterwards, here: python tools/annotate_profiles.py <round tag>; python tools/make_traffic_json.py <summary> <kernel> <points>
# (gpurun MERGES its output into gpurun_out/: remove an earlier session's gpurun_out/prof_<round tag>_* first, or a local re-run of
# tools/parse_profiles.py averages the old CSVs in)
R=${1:?round tag}; HALF=${2:?a or b}
p() { bash tools/profile_gpu.sh "$@" > /dev/null || exit 1; }
if [ "$HALF" == "a" ]; then
  p ${R}_headline
  p ${R}_hosford --workload hosford_update
  p ${R}_hybrid --workload hybrid_update --points 5000000
  p ${R}_ps_update_vjp --workload j2_update_vjp --def-type plane_stress
  p ${R}_objective_grad --workload j2_objective_grad
  p ${R}_barlat --workload j2_update --yield-surface barlat8 --points 2000000
else
  p ${R}_update --workload j2_update
  p ${R}_update_tangent --workload j2_update_tangent
  p ${R}_hill_update_vjp --workload j2_update_vjp --yield-surface hill
  p ${R}_uniaxial_update --workload j2_update --def-type uniaxial_stress --points 2000000
  p ${R}_hosford_update_tangent --workload hosford_update_tangent
  p ${R}_ps_objective_grad --workload j2_objective_grad --def-type plane_stress
fi
ls gpurun_out/prof_${R}_*/summary_*.json
