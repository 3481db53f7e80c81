#!/usr/bin/env bash
# Round 5: merged multi-stream launch on the constant-coefficient kernel (global / unit variances): parity, config-5 times;
# the eager config-3 loop after the host fast paths; the strip kernel's new defaults.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r5_streams
timeout 900 python -m pytest tests -m gpu -q > ${O}_tests.log 2>&1; echo "pytest rc=$?" >> ${O}_tests.log; tail -n 12 ${O}_tests.log | cut -c1-250
timeout 600 python tools/bench_paths.py --only c5 2>&1 | grep '"path"' | tee ${O}_c5.jsonl | cut -c1-330
timeout 300 python tools/dbg/c3_function_overhead.py 2>&1 | grep -v amdgpu.ids | tee ${O}_c3_function_overhead.txt
timeout 300 python tools/bench_paths.py --only c3 2>&1 | grep '"path"' | tee ${O}_c3.jsonl | cut -c1-330
timeout 120 python tools/dbg/strip_variant_time.py all both 2>&1 | grep -v amdgpu.ids | tee ${O}_strip_default.txt
