#!/bin/bash
# round 4: the whole -m gpu suite (no -x: every failure listed)
O=gpurun_out/r04h; mkdir -p $O
SECONDS=0
timeout -k 10 1100 python -m pytest tests -q -m gpu > $O/tests_all.txt 2>&1; echo "all gpu tests rc=$? in ${SECONDS}s"; tail -8 $O/tests_all.txt
