#!/bin/bash
# Runs ON THE GPU BOX: SQ counters of the encoder and decoder kernels on ONE 4096 x 4096 frame (the longest chain / the one
# scan is then the whole kernel, so the per-kernel sums describe the critical path).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-prof_one}
rm -rf $out && mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o one -- python bench.py --frames 1 --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $out/stats.json 2> $out/stats.log
grep "jls::" $out/stats/one_kernel_stats.csv | cut -c1-150 | head -16
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/pmc -o p -- python bench.py --frames 1 --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $out/pmc.json 2> $out/pmc.log
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d $out/pmc2 -o p -- python bench.py --frames 1 --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $out/pmc2.json 2> $out/pmc2.log
find $out -name "*kernel_trace.csv" -size +8M -delete
python tools/summarise_pmc.py $out/pmc | head -14
python tools/summarise_pmc.py $out/pmc2 | head -14
