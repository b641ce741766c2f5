#!/bin/bash
# A/B of cm_objective_grad library variants on the GPU box: values of a seeded ragged batch, then the config-5 bench leg
for L in "" "$@"; do
  echo "${L:-base} values:"; CMAD_HIP_LIB=$L python tools/debug/objective_value.py 3000017
done
bash tools/ab_variants.sh "--workload j2_objective_grad --steps 20 --warmup 5" "$@"
