#!/bin/bash
# Does the ROCr scratch configuration limit how many spilling waves run at once?  (Atlas-30 dID, K = 32768 and 65536)
run() { echo "== $*"; env "$@" timeout -k 10 120 python tools/thread_sweep.py atlas30 3 32768 65536 2>&1 | grep -v amdgpu.ids; }
run X=1
run HSA_NO_SCRATCH_THREAD_LIMITER=1
run HSA_SCRATCH_SINGLE_LIMIT=4000000000
run HSA_SCRATCH_SINGLE_LIMIT=4000000000 HSA_NO_SCRATCH_THREAD_LIMITER=1
run HSA_ENABLE_SCRATCH_ALT=1
run HSA_SCRATCH_SINGLE_LIMIT_ASYNC=4000000000
run HSA_NO_SCRATCH_RECLAIM=1
