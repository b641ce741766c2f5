#!/bin/bash
# A/B of cm_objective_grad library variants on the GPU box: values of a seeded ragged batch, then the config-5 bench leg
cp cmad_amd/csrc/libcmad_hip.so /tmp/base.so
for L in /tmp/base.so "$@"; do
  cp $L cmad_amd/csrc/libcmad_hip.so
  echo "$L values:"; python tools/debug/objective_value.py 3000017
done
cp /tmp/base.so cmad_amd/csrc/libcmad_hip.so
bash tools/ab_variants.sh "--workload j2_objective_grad --steps 20 --warmup 5" "$@"
