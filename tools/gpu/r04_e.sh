#!/bin/bash
# round 4, lease 5: the whole -m gpu suite (timing), smoke, benches after the k_screen / PS fixes
set -o pipefail
O=gpurun_out/r04e; mkdir -p $O
SECONDS=0
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/tests_all.txt 2>&1; echo "all gpu tests rc=$? in ${SECONDS}s"; tail -5 $O/tests_all.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.txt
b() { python bench.py --no-cpu-baseline "$@" 2>>$O/bench.err | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); s=r.get('sustained',{}); print('$*', '| %.4g /s' % r['value'], '| kernel_ms %.4f' % r['roofline']['kernel_ms'], '| sustained ms', s.get('launches_50_250_ms'), 'frac', s.get('sustained_frac'))"; }
b --workload hybrid_update --points 5000000 --sustain
CM_DEBUG_NO_SCREEN=1 python bench.py --no-cpu-baseline --workload hybrid_update --points 5000000 --sustain 2>>$O/bench.err | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); s=r['sustained']; print('hybrid_update NO_SCREEN (pool) sustained ms', s['launches_50_250_ms'])"
b --workload hybrid_update_vjp --points 5000000 --sustain
b --workload j2_update --yield-surface barlat8 --points 2000000 --sustain
b --workload j2_update_vjp --def-type plane_stress --sustain
b --workload hosford_update --sustain
