#!/bin/bash
# occupancy sensitivity: extra dynamic LDS per block limits resident blocks per CU (debug env var)
for l in 0 50000 90000; do
  CM_DEBUG_DYN_LDS=$l python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('dyn_lds', $l, '%.4g' % r['value'], 'kernel_ms', round(r['roofline']['kernel_ms'],4))"
done
