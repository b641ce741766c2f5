#!/bin/bash
# A/B of library builds over the J2 workloads with and without line search: tools/ab_ls.sh lib1.so ...
for WL in "--workload j2_update_vjp" "--workload j2_update_vjp --ls-evals 4" "--workload j2_update" "--workload j2_update --ls-evals 4" "--workload j2_objective_grad" "--workload j2_objective_grad --ls-evals 4"; do
  echo "== $WL"
  bash tools/ab_multi.sh "$WL --steps 10" "$@"
done
