#!/usr/bin/env bash
# Round-2 evidence: rocprofv3 kernel traces (+ stats) and PMC passes of the commands DESIGN.md / README quote.
# Summaries are written next to the raw output as gpurun_out/r02_*.txt (tools/rocpd_summary.py); copy them to profiles/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
sum() { f=$(find gpurun_out/$1 -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_summary.py "$f" $2 > gpurun_out/$1.txt 2>&1; rm -rf gpurun_out/$1; }
# 1. the driver's command form (default kernel choice), with the CPU baseline leg
rocprofv3 --kernel-trace --stats -d gpurun_out/r02_bench_default -o run -- python bench.py --steps 20 --warmup 3 > gpurun_out/r02_bench_default.log 2>&1
grep "^{" gpurun_out/r02_bench_default.log | tail -1 > gpurun_out/r02_bench_default.json; sum r02_bench_default
# 2. the wave-per-system kernel (round 1's, still the choice for narrow streams) on the same workload
rocprofv3 --kernel-trace --stats -d gpurun_out/r02_bench_wave -o run -- python bench.py --steps 20 --warmup 3 --algo 2 --no-cpu-baseline > gpurun_out/r02_bench_wave.log 2>&1
grep "^{" gpurun_out/r02_bench_wave.log | tail -1 > gpurun_out/r02_bench_wave.json; sum r02_bench_wave
# 3. PMC passes (own runs, kernel-trace only), default (= strip) kernel and wave kernel
for a in 0 2; do
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    tag=r02_pmc_algo${a}_$(echo $c | tr ' ' '_')
    rocprofv3 --kernel-trace --pmc $c -d gpurun_out/$tag -o run -- python bench.py --no-cpu-baseline --steps 5 --warmup 1 --algo $a > gpurun_out/$tag.log 2>&1
    sum $tag --pmc
  done
done
# 4. every secondary path
rocprofv3 --kernel-trace --stats -d gpurun_out/r02_paths -o run -- python tools/bench_paths.py > gpurun_out/r02_paths.log 2>&1
grep '"path"' gpurun_out/r02_paths.log > gpurun_out/r02_paths.jsonl; sum r02_paths
# 5. the AUTO policy's data
python tools/algo_sweep.py > gpurun_out/r02_algo_sweep.txt 2>&1
ls -la gpurun_out/r02_* | head -40
cut -c1-220 gpurun_out/r02_paths.jsonl
