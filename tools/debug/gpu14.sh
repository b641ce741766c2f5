#!/bin/bash
# pool chunk size: tail imbalance of the static chunk assignment (10^7 points / 256 = 39062 chunks over 2048 wavefronts = 19.07 each)
for sh in 8 7 6 5; do
  for w in "--workload hosford_update" "--workload hybrid_update --points 5000000"; do
    CM_POOL_CHUNK_SHIFT=$sh CMAD_HIP_LIB=ab_libs/chunk.so python bench.py $w --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('chunk 2^$sh', '$w', '| %.4g' % r['value'], '| kernel_ms %.4f' % r['roofline']['kernel_ms'])"
  done
done
for sh in 8 6 5; do
  for w in "--workload hosford_update" "--workload hybrid_update --points 5000000"; do
    CM_POOL_CHUNK_SHIFT=$sh CMAD_HIP_LIB=ab_libs/chunk.so python bench.py $w --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('chunk 2^$sh', '$w', '| %.4g' % r['value'], '| kernel_ms %.4f' % r['roofline']['kernel_ms'])"
  done
done
