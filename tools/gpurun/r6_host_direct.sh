#!/bin/bash
# round 6: the short host path with arrays >= 1.2 MB copied by the runtime straight from / to the caller's pageable memory
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
O=gpurun_out/host_direct_ab.txt
: > $O
MLPG_HIP_HOST_DIRECT_KB=0 python tools/dbg/host_sizes.py "staged (before)" big >> $O 2>&1
python tools/dbg/host_sizes.py "direct >= 1200 KB" big >> $O 2>&1
MLPG_HIP_HOST_SMALL_MB=100000 python tools/dbg/host_sizes.py "direct >= 1200 KB, everything through the one-stream path" big >> $O 2>&1
MLPG_HIP_HOST_DIRECT_KB=600 python tools/dbg/host_sizes.py "direct >= 600 KB" >> $O 2>&1
MLPG_HIP_HOST_DIRECT_KB=1200 MLPG_HIP_HOST_HELPERS=0 python tools/dbg/host_sizes.py "direct >= 1200 KB, no helper threads" >> $O 2>&1
cat $O
timeout 600 python -m pytest tests/test_literal_calls_gpu.py tests/test_host_multi_gpu.py -x -q 2>&1 | tail -5
