#!/bin/bash
# A/B library builds on the GPU box: tools/ab_variants.sh "<bench args>" variant1.so variant2.so ...   (base = the in-tree library)
ARGS=$1; shift
cp cmad_amd/csrc/libcmad_hip.so /tmp/base.so
for rep in 1 2; do
  for L in /tmp/base.so "$@"; do
    cp $L cmad_amd/csrc/libcmad_hip.so
    python bench.py --no-cpu-baseline $ARGS 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('$L', '$ARGS', '| %.4g' % r['value'], '| ms_per_step %.4f' % r['ms_per_step'], '| kernel_ms %.4f' % r['roofline']['kernel_ms'])"
  done
done
cp /tmp/base.so cmad_amd/csrc/libcmad_hip.so
