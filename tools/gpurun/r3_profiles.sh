#!/usr/bin/env bash
# Round-3 evidence: rocprofv3 kernel traces (+ stats) and PMC passes of the commands DESIGN.md / README quote.
# Summaries land in gpurun_out/r03_*.txt (tools/rocpd_summary.py); copy them to profiles/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
sum() { f=$(find gpurun_out/$1 -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_summary.py "$f" $2 > gpurun_out/$1.txt 2>&1; rm -rf gpurun_out/$1; }
# 1. the driver's command form (default kernel choice), with the CPU baseline leg and the secondary object
rocprofv3 --kernel-trace --stats -d gpurun_out/r03_bench_default -o run -- python bench.py --steps 20 --warmup 3 > gpurun_out/r03_bench_default.log 2>&1
grep "^{" gpurun_out/r03_bench_default.log | tail -1 > gpurun_out/r03_bench_default.json; sum r03_bench_default
# 2. the pipelined kernel on the same workload
rocprofv3 --kernel-trace --stats -d gpurun_out/r03_bench_pipe -o run -- python bench.py --steps 20 --warmup 3 --algo 4 --no-cpu-baseline --no-secondary > gpurun_out/r03_bench_pipe.log 2>&1
grep "^{" gpurun_out/r03_bench_pipe.log | tail -1 > gpurun_out/r03_bench_pipe.json; sum r03_bench_pipe
# 3. PMC passes (own runs, kernel-trace only): shipped strip kernel (algo 0), pipelined kernel (algo 4)
for a in 0 4; do
  for c in FETCH_SIZE WRITE_SIZE; do
    tag=r03_pmc_algo${a}_$c
    rocprofv3 --kernel-trace --pmc $c -d gpurun_out/$tag -o run -- python bench.py --no-cpu-baseline --no-secondary --regions 0 --steps 5 --warmup 1 --algo $a > gpurun_out/$tag.log 2>&1
    sum $tag --pmc
  done
done
# 4. the strip kernel with the halo handed over through LDS (-DMLPG_STRIP_STREAM=2): traffic
MLPG_HIP_EXTRA_FLAGS="-DMLPG_STRIP_STREAM=2" python nnmnkwii_amd/csrc/build.py --quiet --only=mlpg_strip_fwd_f64 > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  tag=r03_pmc_striphalo_$c
  rocprofv3 --kernel-trace --pmc $c -d gpurun_out/$tag -o run -- python bench.py --no-cpu-baseline --no-secondary --regions 0 --steps 5 --warmup 1 --algo 3 > gpurun_out/$tag.log 2>&1
  sum $tag --pmc
done
python nnmnkwii_amd/csrc/build.py --quiet --only=mlpg_strip_fwd_f64 > /dev/null 2>&1
# 5. every secondary path
rocprofv3 --kernel-trace --stats -d gpurun_out/r03_paths -o run -- python tools/bench_paths.py > gpurun_out/r03_paths.log 2>&1
grep '"path"' gpurun_out/r03_paths.log > gpurun_out/r03_paths.jsonl; sum r03_paths
ls -la gpurun_out/r03_* | head -40
cut -c1-200 gpurun_out/r03_paths.jsonl
