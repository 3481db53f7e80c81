#!/usr/bin/env bash
# config 3 through tools/bench_paths.py under rocprofv3: every config-3 kernel's duration (wave, strip, FIR, both training steps)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c3prof -o t -- python tools/bench_paths.py --only c3 --quick > gpurun_out/c3_paths.log 2>&1
echo "rc=$?"
grep '^{' gpurun_out/c3_paths.log | cut -c1-330
f=$(find gpurun_out/c3prof -name 't_kernel_stats.csv' | head -1)
[ -n "$f" ] && grep "mlpg::" "$f" | cut -c1-200
