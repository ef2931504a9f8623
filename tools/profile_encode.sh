#!/bin/bash
# Runs ON THE GPU BOX: kernel statistics and SQ counters of the encoder pipeline on a batch that takes more than one pass.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-prof_enc}
frames=${2:-456}
rm -rf $out && mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o enc -- python bench.py --frames $frames --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $out/stats.json 2> $out/stats.log
find $out/stats -name "*kernel_stats.csv" | head -1 | xargs -r head -20
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/pmc -o p -- python bench.py --frames 64 --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $out/pmc.json 2> $out/pmc.log
find $out -name "*kernel_trace.csv" -size +8M -delete
python tools/summarise_pmc.py $out/pmc | head -40
