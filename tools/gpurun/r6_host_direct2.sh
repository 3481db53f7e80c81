#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
O=gpurun_out/host_direct_ab2.txt
: > $O
MLPG_HIP_HOST_DIRECT_KB=0 python tools/dbg/host_sizes.py "staged, one-stream path up to 64 MB" >> $O 2>&1
python tools/dbg/host_sizes.py "direct >= 1200 KB, one-stream path up to 64 MB" >> $O 2>&1
MLPG_HIP_HOST_DIRECT_KB=0 MLPG_HIP_HOST_SMALL_MB=6 python tools/dbg/host_sizes.py "staged, one-stream path up to 6 MB (the state before)" >> $O 2>&1
cat $O
