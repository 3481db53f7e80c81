#!/usr/bin/env bash
# rocprofv3 kernel-trace summary of the bench command (usage: gpurun_prof.sh <tag> [bench args])
tag=${1:-prof}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d gpurun_out/$tag -o run -- python bench.py --no-cpu-baseline "$@" > gpurun_out/$tag.log 2>&1
tail -2 gpurun_out/$tag.log
find gpurun_out/$tag -name "*stats*" | head; 
f=$(find gpurun_out/$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f"
