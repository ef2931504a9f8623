#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for cfg in "256 2048" "128 1024" "64 512" "64 256" "32 256" "32 128" "128 512"; do
  set -- $cfg
  echo "job $1 warm $2: $(CHARLS_AMD_RUN_JOB_EVENTS=$1 CHARLS_AMD_RUN_WARM_EVENTS=$2 python tools/one_frame_latency.py --calls 10 2>/dev/null | tail -1)"
done
python - <<'PY'
import os, sys, ctypes as C
sys.path.insert(0, os.getcwd())
from charls_amd import capi, synth
import numpy as np
for job, warm in [(256,2048),(64,512),(64,256),(32,128)]:
    os.environ["CHARLS_AMD_RUN_JOB_EVENTS"]=str(job); os.environ["CHARLS_AMD_RUN_WARM_EVENTS"]=str(warm)
    lib = capi.load_product()
    for kind in ["mixed","noise","hard","runs"]:
        try:
            img = synth.frame_numpy(4096, 4096, seed=2, bits=8, kind=kind)
        except Exception as e:
            continue
        raw = C.CDLL(capi.PRODUCT_LIB if hasattr(capi,'PRODUCT_LIB') and capi.PRODUCT_LIB else "charls_amd/lib/libcharls_amd.so")
        before=(C.c_uint64*4)(); raw.charls_amd_speculation_counters(before,4)
        lib.encode(img, width=4096, height=4096, bits_per_sample=8)
        after=(C.c_uint64*4)(); raw.charls_amd_speculation_counters(after,4)
        print(job, warm, kind, [int(after[i]-before[i]) for i in range(4)])
PY
