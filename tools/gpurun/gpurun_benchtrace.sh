#!/usr/bin/env bash
# kernel trace of exactly `python bench.py` (default flags), the command the driver runs
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d gpurun_out/r1f_bench_default -o run -- python bench.py > gpurun_out/r1f_bench_default.log 2>&1
tail -1 gpurun_out/r1f_bench_default.log | cut -c1-300
