#!/bin/bash
# round 4, lease 2: Hosford warm start + private pool counters -- parity tests, pool tests, bench
set -o pipefail
O=gpurun_out/r04b; mkdir -p $O
python -m pytest tests/test_gpu_update.py -x -q -m gpu -k "hosford or work_pool or legacy or random_materials or golden" > $O/tests_update.txt 2>&1; echo "update tests rc=$?"; tail -3 $O/tests_update.txt
python -m pytest tests/test_gpu_pool.py -x -q -m gpu > $O/tests_pool.txt 2>&1; echo "pool tests rc=$?"; tail -5 $O/tests_pool.txt
b() { python bench.py --no-cpu-baseline "$@" 2>>$O/bench.err | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('$*', '| %.4g /s' % r['value'], '| kernel_ms %.4f' % r['roofline']['kernel_ms'], '| frac %.3f' % r['roofline']['frac'], '| sustained', r.get('sustained',{}).get('launches_50_250_ms'))"; }
b --workload hosford_update --steps 10
b --workload hosford_update --steps 10 --general-newton
b --workload hosford_update --sustain
b --workload hosford_update --sustain --general-newton
b --workload hosford_update_vjp --steps 10
b --workload hosford_update_vjp --steps 10 --general-newton
b --workload hosford_update_tangent --steps 10
b --workload hybrid_update --points 5000000 --steps 10
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_hosford -- python3 $R/bench.py --workload hosford_update --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2> $R/$O/prof_hosford.err
cd $R
python3 - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/r04b/prof_hosford/*/*kernel_stats.csv'):
    for r in list(csv.DictReader(open(f)))[:6]:
        print(r['Name'][:110], r['Calls'], r['AverageNs'], r['Percentage'])
PY
