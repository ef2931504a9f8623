#!/bin/bash
# Runs ON THE GPU BOX: PMC passes for the dominant kernel in the instantiation the bench line runs
# (decode_scans_group<uchar, 8, 1>: eight scans per wavefront).  rocprofv3 --pmc crashed inside torch's frame-synthesis
# kernels with 512 / 4096 frames (round 3), so (1) the counters are restricted to this library's kernels, and (2) the same
# instantiation is also measured with 64 frames (CHARLS_AMD_DECODE_GROUP=8 fixes the lanes per scan; bytes and
# instructions per sample do not depend on the number of wavefronts).  Also re-takes the 64-frame traffic of the encoder.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-final}
mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $out/pmc_$c $out/pmc_g8_$c $out/pmc4096_$c
  timeout 240 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "jls" --output-format csv -d $out/pmc_$c -o p -- python bench.py --frames 64 --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $out/pmc_$c.log 2>&1
  CHARLS_AMD_DECODE_GROUP=8 timeout 240 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "decode_scans" --output-format csv -d $out/pmc_g8_$c -o p -- python bench.py --frames 64 --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $out/pmc_g8_$c.log 2>&1
  timeout 500 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "decode_scans" --output-format csv -d $out/pmc4096_$c -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $out/pmc4096_$c.log 2>&1
done
rm -rf $out/pmc_inst_g8
CHARLS_AMD_DECODE_GROUP=8 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-include-regex "decode_scans" --output-format csv -d $out/pmc_inst_g8 -o p -- python bench.py --frames 64 --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $out/pmc_inst_g8.log 2>&1
find $out -name "*kernel_trace.csv" -size +8M -delete
find $out -name "p_counter_collection.csv" | xargs wc -l
