#!/bin/bash
# round 4, lease 4: screened route under capture (fixed reset), warm starts in the near-linear variable, profiles
set -o pipefail
O=gpurun_out/r04d; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_pool.py -x -q -m gpu -k "screened_route_under_graph_capture" > $O/tests_capture.txt 2>&1; rc=$?; echo "capture tests rc=$rc"; tail -3 $O/tests_capture.txt
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_pool.py -x -q -m gpu > $O/tests_pool.txt 2>&1; echo "pool tests rc=$?"; tail -3 $O/tests_pool.txt
timeout -k 10 900 python -m pytest tests/test_gpu_update.py -x -q -m gpu -k "warm or hosford or radial or screened or full_size_properties_plane" > $O/tests_update.txt 2>&1; echo "update tests rc=$?"; tail -3 $O/tests_update.txt
b() { python bench.py --no-cpu-baseline "$@" 2>>$O/bench.err | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); s=r.get('sustained',{}); print('$*', '| %.4g /s' % r['value'], '| kernel_ms %.4f' % r['roofline']['kernel_ms'], '| sustained ms', s.get('launches_50_250_ms'), 'frac', s.get('sustained_frac'))"; }
b --workload hosford_update --sustain
b --workload j2_update_vjp --yield-surface hill --sustain
b --workload j2_update_vjp --yield-surface hill --sustain --reference-iterates
b --workload j2_update_vjp --def-type plane_stress --sustain
b --workload j2_update_vjp --def-type plane_stress --sustain --reference-iterates
b --workload j2_objective_grad --def-type plane_stress --sustain
b --workload j2_objective_grad --def-type plane_stress --sustain --reference-iterates
bash tools/profile_gpu.sh r04_hosford --workload hosford_update > $O/prof_hosford.txt 2>&1; echo "prof hosford rc=$?"
bash tools/profile_gpu.sh r04_hybrid --workload hybrid_update --points 5000000 > $O/prof_hybrid.txt 2>&1; echo "prof hybrid rc=$?"
bash tools/profile_gpu.sh r04_barlat --workload j2_update --yield-surface barlat8 --points 2000000 > $O/prof_barlat.txt 2>&1; echo "prof barlat rc=$?"
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/prof_r04_*/summary_r04_*.json')):
    s=json.load(open(f)); print(f)
    for k,v in s['kernels'].items():
        print('  ',k[:90],'calls',v['calls'],'avg_us %.1f'%v['avg_us'],'vgpr',v.get('VGPR_Count'),'scratch',v.get('Scratch_Size'))
    for k,v in s['pmc_sq'].items():
        print('   SQ',k[:60],{c:round(x['mean']) for c,x in v.items()})
    for k,v in s['pmc_mix'].items():
        print('   MIX',k[:60],{c:round(x['mean']) for c,x in v.items()})
    for nm in ('pmc_fetch','pmc_write'):
        for k,v in s[nm].items():
            print('  ',nm,k[:60],{c:round(x['mean']) for c,x in v.items()})
PY
