#!/usr/bin/env bash
# Round 4, first pass of the constant-coefficient kernel: its parity tests, then timings against wave / strip.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_const_gpu.py -m gpu -x -q > gpurun_out/r4_const1_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r4_const1_tests.log
tail -25 gpurun_out/r4_const1_tests.log
timeout 300 python tools/dbg/const_time.py > gpurun_out/r4_const1_time.log 2>&1
echo "time rc=$?"
cat gpurun_out/r4_const1_time.log
