#!/usr/bin/env bash
# Round 5, after the transposed form: AUTO's routing table regenerated, the whole suite, every secondary path, the narrow-stream table.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python tools/auto_routing.py > gpurun_out/r5_auto_routing.log 2>&1; tail -n 32 gpurun_out/r5_auto_routing.log | cut -c1-260
cp gpurun_out/auto_routing.json profiles/r05_auto_routing.json
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r05_tests_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05_tests_gpu.log; tail -n 8 gpurun_out/r05_tests_gpu.log
timeout 300 python tools/dbg/narrow_time.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_tr_narrow.txt; tail -n 5 gpurun_out/r05_tr_narrow.txt
timeout 600 python tools/bench_paths.py > gpurun_out/r05_paths_untraced.log 2>&1
grep '"path"' gpurun_out/r05_paths_untraced.log > gpurun_out/r05_paths_untraced.jsonl; grep "c5" gpurun_out/r05_paths_untraced.jsonl | cut -c1-160
timeout 200 python tools/dbg/mlpg_soak.py 60 77 2>&1 | tail -n 2
