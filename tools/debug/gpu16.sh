#!/bin/bash
for P in 2000000 1000000 500000; do
 for rep in 1 2; do for L in ab_libs/dmin2048.so ab_libs/dmin512.so; do
  for w in "--workload hybrid_update" "--workload hosford_update"; do
  CMAD_HIP_LIB=$L python bench.py --no-cpu-baseline $w --points $P --steps 10 --warmup 3 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('$L', '$w', $P, '| %.4g' % r['value'], '| kernel_ms %.4f' % r['roofline']['kernel_ms'])"; done; done; done; done
