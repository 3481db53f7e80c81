#!/usr/bin/env bash
# Round 6, final head: the suite, smoke(), the literal-call soak, the other soaks, the bench line (driver form), rocprofv3 stats + PMC passes of the headline kernel
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
sum() { f=$(find gpurun_out/$1 -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_summary.py "$f" $2 > gpurun_out/$1.txt 2>&1; rm -rf gpurun_out/$1; }
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r06_tests_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r06_tests_gpu.log; tail -4 gpurun_out/r06_tests_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 150 python tools/dbg/lit_soak.py 90 601 > gpurun_out/r06_soak_lit.log 2>&1; echo "rc=$?" >> gpurun_out/r06_soak_lit.log; tail -n 3 gpurun_out/r06_soak_lit.log
timeout 200 python tools/dbg/soak_strip.py 90 602 > gpurun_out/r06_soak_strip.log 2>&1; echo "rc=$?" >> gpurun_out/r06_soak_strip.log; tail -n 2 gpurun_out/r06_soak_strip.log
timeout 200 python tools/dbg/mlpg_algos_soak.py 90 603 > gpurun_out/r06_soak_mlpg_algos.log 2>&1; echo "rc=$?" >> gpurun_out/r06_soak_mlpg_algos.log; tail -n 2 gpurun_out/r06_soak_mlpg_algos.log
timeout 200 python tools/dbg/mlpg_soak.py 60 604 > gpurun_out/r06_soak_streams.log 2>&1; echo "rc=$?" >> gpurun_out/r06_soak_streams.log; tail -n 2 gpurun_out/r06_soak_streams.log
timeout 200 python tools/dbg/fir_soak.py 40 605 > gpurun_out/r06_soak_fir.log 2>&1; echo "rc=$?" >> gpurun_out/r06_soak_fir.log; tail -n 2 gpurun_out/r06_soak_fir.log
timeout 200 python tools/dbg/dtw_soak.py 60 606 > gpurun_out/r06_soak_dtw.log 2>&1; echo "rc=$?" >> gpurun_out/r06_soak_dtw.log; tail -n 2 gpurun_out/r06_soak_dtw.log
timeout 200 python tools/dbg/align_soak.py 30 607 > gpurun_out/r06_soak_align.log 2>&1; echo "rc=$?" >> gpurun_out/r06_soak_align.log; tail -n 2 gpurun_out/r06_soak_align.log
echo "== bench (driver form)"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_head.json 2>gpurun_out/r06_bench_head.err; echo "rc=$?"
cut -c1-700 gpurun_out/r06_bench_head.json
echo "== rocprofv3 stats of the metric"
rocprofv3 --kernel-trace --stats -d gpurun_out/r06_bench_metric_head -o run -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/r06_bench_metric_head.log 2>&1
grep "^{" gpurun_out/r06_bench_metric_head.log | tail -1 > gpurun_out/r06_bench_metric_head.json; sum r06_bench_metric_head
for c in FETCH_SIZE WRITE_SIZE; do
  tag=r06_pmc_strip_head_$c
  rocprofv3 --kernel-trace --pmc $c -d gpurun_out/$tag -o run -- python bench.py --no-cpu-baseline --no-secondary --no-traffic --regions 0 --precondition 0 --steps 5 --warmup 1 > gpurun_out/$tag.log 2>&1
  sum $tag --pmc
done
head -5 gpurun_out/r06_bench_metric_head.txt | cut -c1-230
head -4 gpurun_out/r06_pmc_strip_head_FETCH_SIZE.txt | cut -c1-200
head -4 gpurun_out/r06_pmc_strip_head_WRITE_SIZE.txt | cut -c1-200
echo "== every secondary path under the kernel trace"
rocprofv3 --kernel-trace --stats -d gpurun_out/r06_paths_trace -o run -- python tools/bench_paths.py --only c2k,c2g,c2b,c3,c4,c5,ms > gpurun_out/r06_paths.jsonl 2>gpurun_out/r06_paths.err
sum r06_paths_trace; mv gpurun_out/r06_paths_trace.txt gpurun_out/r06_paths.txt 2>/dev/null
head -30 gpurun_out/r06_paths.txt | cut -c1-200
