#!/usr/bin/env bash
# kernel-trace summaries of bench.py and tools/bench_paths.py (final round-1 state)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r1_final_bench -o run -- python bench.py --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/prof_r1_final_bench.log 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r1_final_paths -o run -- python tools/bench_paths.py --quick > gpurun_out/prof_r1_final_paths.log 2>&1
tail -1 gpurun_out/prof_r1_final_bench.log | cut -c1-200
grep path gpurun_out/prof_r1_final_paths.log | cut -c1-160
