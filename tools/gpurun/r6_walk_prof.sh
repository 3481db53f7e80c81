#!/usr/bin/env bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
MLPG_STRIP_WALK=1 rocprofv3 --kernel-trace --stats -d gpurun_out/r6_walk_prof -o run -- python tools/dbg/strip_variant_time.py fwd both > gpurun_out/r6_walk_prof.log 2>&1
f=$(find gpurun_out/r6_walk_prof -name "*.db" | head -1); python tools/rocpd_summary.py "$f" > gpurun_out/r6_walk_prof.txt 2>&1; rm -rf gpurun_out/r6_walk_prof
head -12 gpurun_out/r6_walk_prof.txt | cut -c1-250
