#!/bin/bash
# Round 5, fourth GPU call: wavefronts per workgroup of the group decoder (one wavefront per SIMD by construction), call traces of the thread pool.
out=gpurun_out/r5d
mkdir -p $out
export TMPDIR=/tmp
( timeout 900 python tools/decode_sweep.py --frames 4096 --distinct 64 --groups 8,16:4,32:8,32:4,16,8 --sizes 4096 --repeat 1 ) > $out/decode_wg_4096.txt 2> $out/decode_wg_4096.err
( timeout 600 python tools/decode_sweep.py --frames 2048 --distinct 64 --groups 16,32:4,32:8,16:4 --sizes 1024,2048 --repeat 1 ) > $out/decode_wg_2048.txt 2> $out/decode_wg_2048.err
( CHARLS_AMD_TRACE=1 timeout 600 python tools/threads_abi_probe.py --threads 256 --seconds 6 ) > $out/threads_trace.txt 2> $out/threads_trace.err
( timeout 300 python -m pytest tests/test_gpu_parity.py -q -x --timeout 300 2>&1 | tail -5 ) > $out/pytest_parity.log 2>&1
cat $out/decode_wg_4096.txt $out/decode_wg_2048.txt; cat $out/threads_trace.txt; tail -3 $out/pytest_parity.log
grep -c "charls_amd trace" $out/threads_trace.err
