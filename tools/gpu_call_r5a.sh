#!/bin/bash
# Round 5, first GPU call: the threading contract, the knob migration, clock / power telemetry of the decoder, the bench line.
out=gpurun_out/r5a
mkdir -p $out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_threads.py -q --timeout 300 -x 2>&1 | tail -30 ) > $out/pytest_threads.log 2>&1
( timeout 900 python -m pytest tests -m gpu -q --timeout 600 --deselect tests/test_gpu_threads.py 2>&1 | tail -40 ) > $out/pytest_gpu.log 2>&1
( timeout 600 python tools/decode_clock_power.py --frames 4096 --groups 8,16,32 --repeat 3 ) > $out/decode_clock_power.txt 2> $out/decode_clock_power.err
( timeout 300 python tools/decode_clock_power.py --frames 4096 --groups 8,16,32 --repeat 1 --kind mixed ) > $out/decode_clock_power_mixed.txt 2> $out/decode_clock_power_mixed.err
( timeout 1200 python bench.py ) > $out/bench.json 2> $out/bench.err
tail -3 $out/pytest_threads.log $out/pytest_gpu.log
tail -c 1500 $out/bench.json
