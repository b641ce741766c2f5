#!/bin/bash
# A/B of the occupancy knobs over the J2 workloads with and without line search
for WL in "--workload j2_update_vjp" "--workload j2_update_vjp --ls-evals 4" "--workload j2_update" "--workload j2_update --ls-evals 4" "--workload j2_objective_grad" "--workload j2_objective_grad --ls-evals 4" "--workload hosford_update --steps 5"; do
  echo "== $WL"
  bash tools/ab_multi.sh "$WL --steps 10" ab_libs/v1.so ab_libs/v2.so
done
