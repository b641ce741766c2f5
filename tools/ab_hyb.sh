#!/bin/bash
bash tools/ab_multi.sh "--workload hybrid_update --points 5000000 --steps 5" "$@"
bash tools/ab_multi.sh "--workload hosford_update --steps 5" 
