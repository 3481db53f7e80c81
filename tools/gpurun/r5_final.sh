#!/usr/bin/env bash
# Round 5, final head: the whole GPU suite, smoke(), the bench line as the driver runs it.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r05_tests_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05_tests_gpu.log; tail -n 3 gpurun_out/r05_tests_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 2
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_head.log 2>&1
grep "^{" gpurun_out/r05_bench_head.log | tail -1 > gpurun_out/r05_bench_head.json
cut -c1-400 gpurun_out/r05_bench_head.json
