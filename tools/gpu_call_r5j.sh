#!/bin/bash
out=gpurun_out/final_r5b
mkdir -p $out
export TMPDIR=/tmp
( timeout 1500 python tools/measure_configs.py --only noise,3,5a,5b,5c,5d,5e,5p,6,6w ) > $out/other_configs_rest.txt 2>&1
( timeout 900 python tools/measure_configs.py --only 5a,5b --rgb-frames 1024 ) > $out/other_configs_rgb1024.txt 2>&1
( timeout 1200 python bench.py ) > $out/bench.json 2> $out/bench.err
cat $out/other_configs_rest.txt $out/other_configs_rgb1024.txt; tail -c 400 $out/bench.json
