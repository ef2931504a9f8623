#!/bin/bash
# Round 5, third GPU call: threads again (shared areas kept), effective clock of the decoder by PMC, natural-image data at 4096 frames.
out=gpurun_out/r5c
mkdir -p $out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_threads.py -q --timeout 300 2>&1 | tail -40 ) > $out/pytest_threads.log 2>&1
( timeout 900 python tools/threads_abi_probe.py --threads 256,512 --seconds 12 ) > $out/threads_abi.txt 2> $out/threads_abi.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-include-regex "decode_scans" --output-format csv -d $GRAFT_REPO_ROOT/$out/pmc_clock -o p -- python $GRAFT_REPO_ROOT/tools/decode_sweep.py --frames 4096 --distinct 32 --groups 8,16,32 --sizes 4096 --repeat 0 ) > $out/pmc_clock.log 2>&1
( timeout 400 python tools/decode_clock_power.py --frames 4096 --groups 8,16 --repeat 1 --kind tulips ) > $out/decode_tulips.txt 2> $out/decode_tulips.err
tail -5 $out/pytest_threads.log; cat $out/threads_abi.txt; tail -5 $out/pmc_clock.log; grep "^{" $out/decode_tulips.txt | cut -c1-200
ls $out/pmc_clock | head
