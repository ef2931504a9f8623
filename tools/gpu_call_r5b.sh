#!/bin/bash
# Round 5, second GPU call: coalescer v2 (early announcements, bounded concurrency), telemetry of OUR card, threads_abi alone.
out=gpurun_out/r5b
mkdir -p $out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_threads.py -q --timeout 300 2>&1 | tail -40 ) > $out/pytest_threads.log 2>&1
( timeout 300 python tools/concurrent_kernels_probe.py ) > $out/concurrent_kernels.txt 2> $out/concurrent_kernels.err
( timeout 600 python tools/decode_clock_power.py --frames 4096 --groups 8,16,32 --repeat 2 ) > $out/decode_clock_power.txt 2> $out/decode_clock_power.err
( timeout 900 python tools/threads_abi_probe.py --threads 256,512 --seconds 12 ) > $out/threads_abi.txt 2> $out/threads_abi.err
tail -5 $out/pytest_threads.log; cat $out/concurrent_kernels.txt; tail -5 $out/threads_abi.txt
