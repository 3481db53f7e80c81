#!/usr/bin/env bash
# Round-4 evidence: rocprofv3 kernel traces (+ stats) and PMC passes of the commands DESIGN.md / README / r04_notes quote.
# Summaries land in gpurun_out/r04_*.txt (tools/rocpd_summary.py); copy them to profiles/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
sum() { f=$(find gpurun_out/$1 -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_summary.py "$f" $2 > gpurun_out/$1.txt 2>&1; rm -rf gpurun_out/$1; }
# 1. the driver's command form, with the CPU baseline leg and the secondary object
rocprofv3 --kernel-trace --stats -d gpurun_out/r04_bench_default -o run -- python bench.py --steps 20 --warmup 3 > gpurun_out/r04_bench_default.log 2>&1
grep "^{" gpurun_out/r04_bench_default.log | tail -1 > gpurun_out/r04_bench_default.json; sum r04_bench_default
# 2. the metric alone
rocprofv3 --kernel-trace --stats -d gpurun_out/r04_bench_metric -o run -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/r04_bench_metric.log 2>&1
grep "^{" gpurun_out/r04_bench_metric.log | tail -1 > gpurun_out/r04_bench_metric.json; sum r04_bench_metric
# 3. PMC passes (own runs, kernel-trace only): the headline strip kernel
for c in FETCH_SIZE WRITE_SIZE; do
  tag=r04_pmc_strip_$c
  rocprofv3 --kernel-trace --pmc $c -d gpurun_out/$tag -o run -- python bench.py --no-cpu-baseline --no-secondary --regions 0 --steps 5 --warmup 1 > gpurun_out/$tag.log 2>&1
  sum $tag --pmc
done
# 4. the constant-coefficient kernel (config-2 shape, global variances, float64): kernel trace and PMC, forward and backward
for d in fwd bwd; do
  tag=r04_const_${d}_trace
  rocprofv3 --kernel-trace --stats -d gpurun_out/$tag -o run -- python tools/dbg/const_timing.py 256 1000 60 f64 global $d > gpurun_out/$tag.log 2>&1
  sum $tag
  for c in FETCH_SIZE WRITE_SIZE; do
    tag=r04_pmc_const_${d}_$c
    rocprofv3 --kernel-trace --pmc $c -d gpurun_out/$tag -o run -- python tools/dbg/const_timing.py 256 1000 60 f64 global $d > gpurun_out/$tag.log 2>&1
    sum $tag --pmc
  done
done
# 4b. the chunked kernel (the reference's 5-tap windows at the config-2 shape): kernel trace and PMC per pass
rocprofv3 --kernel-trace --stats -d gpurun_out/r04_chunk_trace -o run -- python tools/dbg/chunk_time.py > gpurun_out/r04_chunk_trace.log 2>&1
sum r04_chunk_trace
for c in FETCH_SIZE WRITE_SIZE; do
  tag=r04_pmc_chunk_$c
  rocprofv3 --kernel-trace --pmc $c -d gpurun_out/$tag -o run -- python tools/dbg/chunk_time.py > gpurun_out/$tag.log 2>&1
  sum $tag --pmc
done
# 5. every secondary path
rocprofv3 --kernel-trace --stats -d gpurun_out/r04_paths -o run -- python tools/bench_paths.py > gpurun_out/r04_paths.log 2>&1
grep '"path"' gpurun_out/r04_paths.log > gpurun_out/r04_paths.jsonl; sum r04_paths
# 6. untraced: the bench line as the driver will see it
python bench.py --steps 20 --warmup 3 > gpurun_out/r04_bench_final.log 2>&1
grep "^{" gpurun_out/r04_bench_final.log | tail -1 > gpurun_out/r04_bench_final.json
ls -la gpurun_out/r04_* | head -40
cut -c1-200 gpurun_out/r04_paths.jsonl
cut -c1-600 gpurun_out/r04_bench_final.json
