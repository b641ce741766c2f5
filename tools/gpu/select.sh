#!/bin/bash
# A short lease for one change: selected GPU tests, then bench lines (one per remaining argument string).
#   gpurun --timeout 900 -- 'bash tools/gpu/select.sh <tag> "<pytest -k expression>" "<bench args>" ["<bench args>" ...]'
TAG=${1:?tag}; K=${2:?pytest -k expression}; shift 2
O=gpurun_out/$TAG; mkdir -p $O
SECONDS=0
timeout -k 10 800 python -m pytest tests -x -q -m gpu -k "$K" > $O/tests.txt 2>&1; rc=$?
echo "tests rc=$rc in ${SECONDS}s"; tail -4 $O/tests.txt
[ $rc -eq 0 ] || exit $rc
for args in "$@"; do
  python bench.py --no-cpu-baseline $args 2>>$O/bench.err | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); s=r.get('sustained',{}); print('$args', '| %.4g /s' % r['value'], '| kernel_ms %.4f' % r['roofline']['kernel_ms'], '| sustained ms', s.get('launches_50_250_ms'), 'frac', s.get('sustained_frac'))" || exit 1
done
