#!/bin/bash
# round 4 profiles, part A: rocprofv3 kernel-trace stats + PMC passes (tools/profile_gpu.sh) of six bench workloads
R=r04
timeout -k 10 300 python -m pytest tests/test_gpu_facade.py -x -q -m gpu -k "extended_parameter_limit or extended_leaves" > gpurun_out/r04_ep_test.txt 2>&1; echo "ep tests rc=$?"; tail -2 gpurun_out/r04_ep_test.txt
timeout -k 10 300 python -m pytest tests/test_gpu_update.py -x -q -m gpu -k "barlat or screened" > gpurun_out/r04_barlat_test.txt 2>&1; echo "barlat tests rc=$?"; tail -2 gpurun_out/r04_barlat_test.txt
bash tools/profile_gpu.sh ${R}_headline > /dev/null || exit 1
bash tools/profile_gpu.sh ${R}_hosford --workload hosford_update > /dev/null || exit 1
bash tools/profile_gpu.sh ${R}_hybrid --workload hybrid_update --points 5000000 > /dev/null || exit 1
bash tools/profile_gpu.sh ${R}_ps_update_vjp --workload j2_update_vjp --def-type plane_stress > /dev/null || exit 1
bash tools/profile_gpu.sh ${R}_objective_grad --workload j2_objective_grad > /dev/null || exit 1
bash tools/profile_gpu.sh ${R}_barlat --workload j2_update --yield-surface barlat8 --points 2000000 > /dev/null || exit 1
ls gpurun_out/prof_${R}_*/summary_*.json
