bash tools/ab_variants.sh "--workload j2_update --yield-surface hill --steps 10 --warmup 3" ab_libs/poolhill.so 2>&1 | tee gpurun_out/r3_pool_hill_ab.txt
bash tools/ab_variants.sh "--workload j2_update --yield-surface hill --def-type plane_stress --steps 10 --warmup 3" ab_libs/poolhill.so 2>&1 | tee -a gpurun_out/r3_pool_hill_ab.txt
