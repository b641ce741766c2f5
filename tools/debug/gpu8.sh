set -o pipefail
python -m pytest tests -m gpu -x -q > gpurun_out/r3_gpu_tests6.log 2>&1; echo "tests rc=$?"
tail -4 gpurun_out/r3_gpu_tests6.log
run() { python bench.py --no-cpu-baseline --steps 10 --warmup 3 "$@" 2>>gpurun_out/r3_bench6.err | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('$*', '| %.4g updates/s' % r['value'], '| ms_per_step %.4f' % r['ms_per_step'], '| kernel_ms %.4f' % r['roofline']['kernel_ms'], '| frac %.3f' % r['roofline']['frac'])"; }
(run --workload hosford_update
run --workload hosford_update --lockstep
run --workload hosford_update_vjp
run --workload hosford_update_tangent
run --workload j2_update_vjp --yield-surface hosford8
run --workload hybrid_update --points 5000000) 2>&1 | tee gpurun_out/r3_bench6.txt
