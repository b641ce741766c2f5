#!/bin/bash
# A/B several library builds on the GPU box: tools/ab_multi.sh "<bench args>" lib1.so lib2.so ...
ARGS=$1; shift
cp cmad_amd/csrc/libcmad_hip.so /tmp/base.so
for rep in 1 2; do
  for L in /tmp/base.so "$@"; do
    cp $L cmad_amd/csrc/libcmad_hip.so
    python bench.py --no-cpu-baseline $ARGS 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('$L', '%.4g' % r['value'], 'kernel_ms', round(r['roofline']['kernel_ms'],4))"
  done
done
cp /tmp/base.so cmad_amd/csrc/libcmad_hip.so
