set -o pipefail
bash tools/ab_variants.sh "--workload hosford_update --steps 5 --warmup 2" ab_libs/refill8.so ab_libs/refill4.so ab_libs/refill32.so 2>&1 | tee gpurun_out/r3_refill_ab.txt
bash tools/ab_variants.sh "--workload hybrid_update --points 5000000 --steps 5 --warmup 2" ab_libs/refill8.so ab_libs/refill4.so ab_libs/refill32.so 2>&1 | tee -a gpurun_out/r3_refill_ab.txt
bash tools/profile_gpu.sh r03a_hosford --workload hosford_update > gpurun_out/prof_r03a_hosford.log 2>&1 && echo prof hosford ok
bash tools/profile_gpu.sh r03a_hybrid --workload hybrid_update --points 5000000 > gpurun_out/prof_r03a_hybrid.log 2>&1 && echo prof hybrid ok
