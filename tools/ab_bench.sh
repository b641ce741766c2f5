#!/bin/bash
# A/B two builds of the library on the GPU box: tools/ab_bench.sh <alt.so> [bench args]
ALT=$1; shift
for i in 1 2; do
  python bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('base', r['value'], r['roofline']['kernel_ms'])"
  cp cmad_amd/csrc/libcmad_hip.so /tmp/base.so; cp $ALT cmad_amd/csrc/libcmad_hip.so
  python bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('alt ', r['value'], r['roofline']['kernel_ms'])"
  cp /tmp/base.so cmad_amd/csrc/libcmad_hip.so
done
