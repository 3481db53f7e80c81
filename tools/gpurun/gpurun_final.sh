#!/usr/bin/env bash
# final round-1 evidence: PMC traffic of the shipped K1 build (separate passes) + kernel trace of the DTW path
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d gpurun_out/r1f_$c -o run -- python bench.py --no-cpu-baseline --steps 5 --warmup 1 > gpurun_out/r1f_$c.log 2>&1
  tail -1 gpurun_out/r1f_$c.log | cut -c1-160
done
rocprofv3 --kernel-trace --stats -d gpurun_out/r1f_dtw -o run -- python tools/bench_paths.py --only c4 > gpurun_out/r1f_dtw.log 2>&1
grep path gpurun_out/r1f_dtw.log | cut -c1-200
