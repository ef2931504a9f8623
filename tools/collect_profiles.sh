#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the bench line, the rocprofv3 kernel statistics of the same command and the PMC
# passes that DESIGN.md / bench.py quote.  Everything lands in gpurun_out/<dir>/; tools/summarise_profiles.py turns it
# into the files committed under profiles/.  The HBM traffic of the encoder is taken at 64 frames (all kernels of the tile
# pipeline), traffic and instruction counts of the dominant kernel -- since round 5 decode_scans_group<uchar, 16, 1, 4> (round 6: its step loop in assembly): four
# scans per wavefront, four wavefronts per workgroup -- with the bench's own 4096 frames (decode is one launch: the counters
# are per launch).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-final}
rm -rf $out && mkdir -p $out
python bench.py > $out/bench.json 2> $out/bench.err
python -m pytest tests -m gpu -q --timeout 600 2>&1 | grep -E "passed|failed|error|FAILED" | tail -8 > $out/pytest_gpu.log
tail -c 600 $out/bench.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o bench -- python bench.py --no-cpu-baseline --no-extras > $out/bench_under_rocprof.json 2> $out/stats.log
rm -f $out/stats/*/bench_kernel_trace.csv $out/stats/bench_kernel_trace.csv
# PMC passes: counters restricted to this library's kernels (rocprofv3 --pmc crashed inside torch's frame-synthesis kernels
# with 512 and 4096 frames in round 3, also with the filter: the dominant kernel's own instantiation is therefore measured
# with 64 frames and its launch shape forced by knobs; bytes and instructions per sample do not depend on the number of
# wavefronts)
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 240 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "jls" --output-format csv -d $out/pmc_$c -o p -- python bench.py --frames 64 --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $out/pmc_$c.log 2>&1
done
timeout 240 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-include-regex "jls" --output-format csv -d $out/pmc_inst -o p -- python bench.py --frames 64 --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $out/pmc_inst.log 2>&1
CHARLS_AMD_DECODE_GROUP=16 CHARLS_AMD_DECODE_WORKGROUP_WAVES=4 timeout 240 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-include-regex "decode_scans" --output-format csv -d $out/pmc_inst_g8 -o p -- python bench.py --frames 64 --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $out/pmc_inst_g8.log 2>&1
# the dominant kernel with the bench's own 4096 frames: 64 distinct frames repeated, so that torch's synthesis is a few kernels
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "decode_scans" --output-format csv -d $out/pmc4096_$c -o p -- python bench.py --distinct 64 --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $out/pmc4096_$c.log 2>&1
done
# one frame through the host-pointer C ABI: latency, and where it goes
python tools/one_frame_latency.py --decode > $out/one_frame_latency.txt 2>&1
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/one -o one -- python tools/one_frame_latency.py --calls 6 > $out/one.log 2>&1
python tools/copy_probe.py > $out/copy_probe.txt 2>&1
# the other BASELINE configurations (parity cases, not bench lines): EVERY row of DESIGN 6.3
timeout 1200 python tools/measure_configs.py --only 2,2b,noise,3,5a,5b,5c,5d,5e,5p,6,6w > $out/other_configs.txt 2>&1
# the headline decoder on a natural image (the reference's tulips, tiled) at the bench's 4096 frames, and BASELINE configs[3] as stated on this one GPU
timeout 400 python tools/decode_clock_power.py --frames 4096 --groups -1 --repeat 1 --kind tulips > $out/decode_tulips.txt 2>&1
timeout 400 python bench.py --workload cfg4 --no-cpu-baseline > $out/bench_cfg4.json 2> $out/bench_cfg4.err
find $out -name "*kernel_trace.csv" -size +8M -delete
du -sh $out; find $out -type f | head -40
