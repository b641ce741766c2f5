set -o pipefail
python -m pytest tests -m gpu -x -q > gpurun_out/r3_gpu_tests5.log 2>&1; echo "tests rc=$?"
tail -3 gpurun_out/r3_gpu_tests5.log
run() { python bench.py --no-cpu-baseline --steps 20 --warmup 5 "$@" 2>>gpurun_out/r3_bench5.err | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('$*', '| %.4g updates/s' % r['value'], '| ms_per_step %.4f' % r['ms_per_step'], '| kernel_ms %.4f' % r['roofline']['kernel_ms'], '| frac %.3f' % r['roofline']['frac'])"; }
(run --workload j2_update_vjp
run --workload j2_objective_grad
run --workload j2_update
run --workload j2_update_tangent
run --workload j2_update_vjp --def-type plane_stress
run --workload j2_update_vjp --yield-surface hill
run --workload j2_update_vjp
python tools/objective_latency.py) 2>&1 | tee gpurun_out/r3_bench5.txt
