#!/bin/bash
# One GPU lease, everything the round-end driver runs plus the sustained table:
#   gpurun --timeout 1190 -- 'bash tools/gpu/suite.sh <tag>'
# -> gpurun_out/<tag>/{tests_all.txt, smoke.txt, bench.json, sustained.txt}.  A failed or timed-out step ends the lease: nothing
# touches the GPU after it.
TAG=${1:-suite}; O=gpurun_out/$TAG; mkdir -p $O
SECONDS=0
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/tests_all.txt 2>&1; rc=$?
echo "all gpu tests rc=$rc in ${SECONDS}s"; tail -4 $O/tests_all.txt
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; rc=$?; echo "smoke rc=$rc"; tail -1 $O/smoke.txt
[ $rc -eq 0 ] || exit $rc
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; rc=$?; echo "bench rc=$rc"
[ $rc -eq 0 ] || exit $rc
[ "$2" == "nosustain" ] && exit 0
tools/bench_sustain.sh > $O/sustained.txt 2>&1
grep -A1 "^--workload" $O/sustained.txt | grep -v "^--$" | cut -c1-200
