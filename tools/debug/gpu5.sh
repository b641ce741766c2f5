bash tools/ab_variants.sh "--workload j2_update_vjp --steps 20 --warmup 5" ab_libs/nohnn.so 2>&1 | tee gpurun_out/r3_hnn_ab.txt
bash tools/ab_variants.sh "--workload j2_objective_grad --steps 20 --warmup 5" ab_libs/nohnn.so 2>&1 | tee -a gpurun_out/r3_hnn_ab.txt
bash tools/ab_variants.sh "--workload j2_update --steps 20 --warmup 5" ab_libs/nohnn.so 2>&1 | tee -a gpurun_out/r3_hnn_ab.txt
bash tools/ab_variants.sh "--workload j2_update_vjp --def-type plane_stress --steps 20 --warmup 5" ab_libs/nohnn.so 2>&1 | tee -a gpurun_out/r3_hnn_ab.txt
