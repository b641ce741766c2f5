#!/bin/bash
# round 4: the whole -m gpu suite on the final library, then the sustained table
O=gpurun_out/r04g; mkdir -p $O
SECONDS=0
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/tests_all.txt 2>&1; echo "all gpu tests rc=$? in ${SECONDS}s"; tail -4 $O/tests_all.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
tools/bench_sustain.sh > $O/sustained.txt 2>&1
grep -A1 "^--workload" $O/sustained.txt | grep -v "^--$" | cut -c1-200
