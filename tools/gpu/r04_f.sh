#!/bin/bash
# round 4: sustained figures on the final library + the driver's default bench command
O=gpurun_out/r04f; mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
tools/bench_sustain.sh > $O/sustained.txt 2>&1
cat $O/sustained.txt
