#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for cfg in "1024 1024" "512 1024" "256 1024" "512 512" "256 512" "128 512"; do
  set -- $cfg
  echo "job $1 warm $2: $(CHARLS_AMD_JOB_EVENTS=$1 CHARLS_AMD_WARM_EVENTS=$2 python tools/one_frame_latency.py --calls 10 2>/dev/null | tail -1)"
done
