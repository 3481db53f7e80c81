#!/usr/bin/env bash
# Round-3 evidence, second pass (after the fastdtw and multi-stream work; the strip kernel itself is unchanged, so the PMC
# passes of r3_profiles.sh stand): kernel traces of the driver's bench command, of the metric alone, of every secondary path.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
sum() { f=$(find gpurun_out/$1 -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_summary.py "$f" $2 > gpurun_out/$1.txt 2>&1; rm -rf gpurun_out/$1; }
rocprofv3 --kernel-trace --stats -d gpurun_out/r03_bench_default -o run -- python bench.py --steps 20 --warmup 3 > gpurun_out/r03_bench_default.log 2>&1
grep "^{" gpurun_out/r03_bench_default.log | tail -1 > gpurun_out/r03_bench_default.json; sum r03_bench_default
rocprofv3 --kernel-trace --stats -d gpurun_out/r03_bench_metric -o run -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/r03_bench_metric.log 2>&1
grep "^{" gpurun_out/r03_bench_metric.log | tail -1 > gpurun_out/r03_bench_metric.json; sum r03_bench_metric
rocprofv3 --kernel-trace --stats -d gpurun_out/r03_paths -o run -- python tools/bench_paths.py > gpurun_out/r03_paths.log 2>&1
grep '"path"' gpurun_out/r03_paths.log > gpurun_out/r03_paths.jsonl; sum r03_paths
# untraced: the bench line as the driver will see it
python bench.py --steps 20 --warmup 3 > gpurun_out/r03_bench_final.log 2>&1
grep "^{" gpurun_out/r03_bench_final.log | tail -1 > gpurun_out/r03_bench_final.json
cut -c1-220 gpurun_out/r03_paths.jsonl
cut -c1-600 gpurun_out/r03_bench_final.json
