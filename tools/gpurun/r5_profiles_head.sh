#!/usr/bin/env bash
# Round 5, final head: rocprofv3 kernel trace + stats of the metric run, PMC passes of the headline kernel (the kernel's template list
# grew by one argument with the transposed form; the kernel itself is unchanged)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
sum() { f=$(find gpurun_out/$1 -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_summary.py "$f" $2 > gpurun_out/$1.txt 2>&1; rm -rf gpurun_out/$1; }
rocprofv3 --kernel-trace --stats -d gpurun_out/r05_bench_metric_head -o run -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/r05_bench_metric_head.log 2>&1
grep "^{" gpurun_out/r05_bench_metric_head.log | tail -1 > gpurun_out/r05_bench_metric_head.json; sum r05_bench_metric_head
for c in FETCH_SIZE WRITE_SIZE; do
  tag=r05_pmc_strip_head_$c
  rocprofv3 --kernel-trace --pmc $c -d gpurun_out/$tag -o run -- python bench.py --no-cpu-baseline --no-secondary --no-traffic --regions 0 --precondition 0 --steps 5 --warmup 1 > gpurun_out/$tag.log 2>&1
  sum $tag --pmc
done
head -4 gpurun_out/r05_bench_metric_head.txt | cut -c1-230
head -4 gpurun_out/r05_pmc_strip_head_FETCH_SIZE.txt | cut -c1-200
head -4 gpurun_out/r05_pmc_strip_head_WRITE_SIZE.txt | cut -c1-200
cut -c1-330 gpurun_out/r05_bench_metric_head.json
