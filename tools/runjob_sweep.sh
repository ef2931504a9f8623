#!/bin/bash
# Runs ON THE GPU BOX: ONE 4096 x 4096 frame through the host-pointer ABI for several sizes of the run chain's jobs and of
# their warm-up (CHARLS_AMD_RUN_JOB_EVENTS / CHARLS_AMD_RUN_WARM_EVENTS): where the defaults for small batches come from
# (profiles/r04_run_job_sweep.txt).  A warm-up that is too short is not an error -- every job is walked again, serially.
cd "$GRAFT_REPO_ROOT"
for cfg in "256 2048" "128 1024" "64 512" "64 256" "32 256" "32 128" "128 512"; do
  set -- $cfg
  echo "job $1 warm $2: $(CHARLS_AMD_RUN_JOB_EVENTS=$1 CHARLS_AMD_RUN_WARM_EVENTS=$2 python tools/one_frame_latency.py --calls 10 2>/dev/null | tail -1)"
done
