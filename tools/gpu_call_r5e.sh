#!/bin/bash
# Round 5, fifth GPU call: the new launch rule of the decoder (one wavefront per SIMD), launch streams with priorities.
out=gpurun_out/r5e
mkdir -p $out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -30 ) > $out/pytest_gpu.log 2>&1
( timeout 300 python tools/concurrent_kernels_probe.py ) > $out/concurrent_kernels.txt 2> $out/concurrent_kernels.err
( timeout 900 python tools/threads_abi_probe.py --threads 256,512 --seconds 12 ) > $out/threads_abi.txt 2> $out/threads_abi.err
( timeout 600 python tools/decode_sweep.py --frames 4096 --distinct 64 --groups -1 --sizes 256,1024,2048,3000,4096 --repeat 1 ) > $out/decode_default_rule.txt 2> $out/decode_default_rule.err
( timeout 400 python tools/decode_clock_power.py --frames 4096 --groups -1 --repeat 1 --kind tulips ) > $out/decode_tulips.txt 2> $out/decode_tulips.err
tail -4 $out/pytest_gpu.log; cat $out/concurrent_kernels.txt $out/threads_abi.txt $out/decode_default_rule.txt; grep "^{" $out/decode_tulips.txt | cut -c1-220
