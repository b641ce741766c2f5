set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r3_gpu_tests1.log 2>&1; echo "tests rc=$?" 
tail -3 gpurun_out/r3_gpu_tests1.log
run() { python bench.py --no-cpu-baseline --steps 5 --warmup 2 "$@" 2>>gpurun_out/r3_bench1.err | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('$*', '| %.4g updates/s' % r['value'], '| ms_per_step %.4f' % r['ms_per_step'], '| kernel_ms %.4f' % r['roofline']['kernel_ms'], '| frac %.3f' % r['roofline']['frac'])"; }
(run --workload hosford_update
run --workload hosford_update --lockstep
run --workload hybrid_update --points 5000000
run --workload hybrid_update --points 5000000 --lockstep
run --workload hosford_update_vjp
run --workload hybrid_update_vjp --points 5000000
run --workload j2_update_vjp --ls-evals 4 --general-newton
run --workload j2_update_vjp --def-type plane_stress --ls-evals 4
run --workload j2_update_vjp) 2>&1 | tee gpurun_out/r3_bench1.txt
