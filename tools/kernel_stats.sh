#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel statistics of one command, the library's kernels first.
#   tools/kernel_stats.sh <name> <command...>   ->  gpurun_out/<name>/stats.csv (+ the command's output in run.log)
set -u
name=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$name
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/raw -o k -- "$@" > $out/run.log 2>&1
f=$(find $out/raw -name "k_kernel_stats.csv" | head -1)
if [ -n "$f" ]; then
  head -1 "$f" > $out/stats.csv
  grep "jls" "$f" | head -40 >> $out/stats.csv
  rm -rf $out/raw
fi
cut -c1-200 $out/stats.csv | head -30
tail -3 $out/run.log
