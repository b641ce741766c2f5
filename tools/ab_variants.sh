#!/bin/bash
# Same-box A/B of library builds on the GPU box: tools/ab_variants.sh "<bench args>" variant1.so variant2.so ...
# (base = the in-tree library).  Variants are selected through CMAD_HIP_LIB (cmad_amd/_lib.py); the in-tree build output is
# never overwritten, so an interrupted run leaves nothing behind.
ARGS=$1; shift
for rep in 1 2; do
  for L in "" "$@"; do
    CMAD_HIP_LIB=$L python bench.py --no-cpu-baseline $ARGS 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('${L:-base}', '$ARGS', '| %.4g' % r['value'], '| ms_per_step %.4f' % r['ms_per_step'], '| kernel_ms %.4f' % r['roofline']['kernel_ms'])"
  done
done
