#!/usr/bin/env bash
# Round-5 evidence: rocprofv3 kernel traces (+ stats) and PMC passes of the commands DESIGN.md / README / r05_notes quote, taken
# AFTER the last kernel change.  Summaries land in gpurun_out/r05_*.txt (tools/rocpd_summary.py); copy them to profiles/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
sum() { f=$(find gpurun_out/$1 -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_summary.py "$f" $2 > gpurun_out/$1.txt 2>&1; rm -rf gpurun_out/$1; }
# 0. the whole suite at this head
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r05_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05_tests.log; tail -n 3 gpurun_out/r05_tests.log
# 1. the driver's command form, with the CPU baseline leg and the secondary object
rocprofv3 --kernel-trace --stats -d gpurun_out/r05_bench_default -o run -- python bench.py --steps 20 --warmup 3 > gpurun_out/r05_bench_default.log 2>&1
grep "^{" gpurun_out/r05_bench_default.log | tail -1 > gpurun_out/r05_bench_default.json; sum r05_bench_default
# 2. the metric alone
rocprofv3 --kernel-trace --stats -d gpurun_out/r05_bench_metric -o run -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/r05_bench_metric.log 2>&1
grep "^{" gpurun_out/r05_bench_metric.log | tail -1 > gpurun_out/r05_bench_metric.json; sum r05_bench_metric
# 3. PMC passes (own runs, kernel-trace only): the headline strip kernel
for c in FETCH_SIZE WRITE_SIZE; do
  tag=r05_pmc_strip_$c
  rocprofv3 --kernel-trace --pmc $c -d gpurun_out/$tag -o run -- python bench.py --no-cpu-baseline --no-secondary --no-traffic --regions 0 --steps 5 --warmup 1 > gpurun_out/$tag.log 2>&1
  sum $tag --pmc
done
# 4. the strip backward (config-2 shape), float64 and float32: kernel trace
for d in f64 f32; do
  tag=r05_strip_bwd_${d}_trace
  rocprofv3 --kernel-trace --stats -d gpurun_out/$tag -o run -- python tools/dbg/strip_variant_time.py bwd $d > gpurun_out/$tag.log 2>&1
  sum $tag
done
# 5. the FIR kernels at 256 x 1000 x 60 and at config 3 (kernel durations)
for shape in "64 500 60" "256 1000 60"; do
  tag=r05_fir_$(echo $shape | tr ' ' x)_trace
  rocprofv3 --kernel-trace --stats -d gpurun_out/$tag -o run -- python tools/dbg/fir_run.py $shape > gpurun_out/$tag.log 2>&1
  sum $tag
done
# 6. every secondary path
rocprofv3 --kernel-trace --stats -d gpurun_out/r05_paths -o run -- python tools/bench_paths.py > gpurun_out/r05_paths.log 2>&1
grep '"path"' gpurun_out/r05_paths.log > gpurun_out/r05_paths.jsonl; sum r05_paths
# 7. untraced: every secondary path, and the bench line as the driver will see it
python tools/bench_paths.py > gpurun_out/r05_paths_untraced.log 2>&1
grep '"path"' gpurun_out/r05_paths_untraced.log > gpurun_out/r05_paths_untraced.jsonl
python bench.py --steps 20 --warmup 3 > gpurun_out/r05_bench_final.log 2>&1
grep "^{" gpurun_out/r05_bench_final.log | tail -1 > gpurun_out/r05_bench_final.json
python tools/dbg/host_path_time.py > gpurun_out/r05_host_path_time.txt 2>&1
ls -la gpurun_out/r05_* | head -40
cut -c1-220 gpurun_out/r05_paths_untraced.jsonl
cut -c1-700 gpurun_out/r05_bench_final.json
